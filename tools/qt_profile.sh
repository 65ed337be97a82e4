# phase timing of k_quadtree on the GPU box: tests first (product build), then a profiling build of the library
set -e
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_extractor.py -x -q -m gpu > gpurun_out/qt_tests.log 2>&1 || { tail -30 gpurun_out/qt_tests.log; exit 1; }
tail -2 gpurun_out/qt_tests.log
ORBX_EXTRA_FLAGS=-DORBX_QT_PROFILE python -m orb_slam3_modified_amd.build --force > gpurun_out/qtprof_build.log 2>&1
timeout 300 python tools/qt_profile.py > gpurun_out/qtprof.txt 2>&1
cat gpurun_out/qtprof.txt
