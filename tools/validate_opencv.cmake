# How a maintainer builds tools/validate_opencv.cpp against the OpenCV ORB_SLAM3 is built with (include from the reference's
# CMakeLists.txt after find_package(OpenCV), or use as a stand-alone project: cmake -DORBX_ROOT=/path/to/this/repo -DORB_SLAM3_ROOT=... -P is
# NOT enough — add_executable needs a project; copy the lines below into a CMakeLists.txt).
#
#   find_package(OpenCV 4.4 REQUIRED)          # or 3.2: whatever ORB_SLAM3 itself uses (CMakeLists.txt:33-39)
#   set(ORBX_ROOT /path/to/orb_slam3_modified_amd-repo)
#   # the oracle (the arithmetic orbx equals), built once with its own Makefile:  make -C ${ORBX_ROOT}/oracle
#   add_executable(validate_opencv ${ORBX_ROOT}/tools/validate_opencv.cpp
#                  ${PROJECT_SOURCE_DIR}/src/ORBextractor.cc)                 # the reference's own extractor, for the operator() leg
#   target_compile_definitions(validate_opencv PRIVATE ORBX_VALIDATE_REFERENCE ORBX_VALIDATE_EXTRAS)
#   target_include_directories(validate_opencv PRIVATE ${PROJECT_SOURCE_DIR}/include   # FIRST: "ORBextractor.h" must be the reference's
#                              ${ORBX_ROOT}/include ${OpenCV_INCLUDE_DIRS})
#   target_link_directories(validate_opencv PRIVATE ${ORBX_ROOT}/oracle ${ORBX_ROOT}/orb_slam3_modified_amd)
#   target_link_libraries(validate_opencv ${OpenCV_LIBS} orb_oracle orbx)
#   # use the flags ORB_SLAM3 is compiled with (-O3 -march=native): the brief_fma detection looks at THIS build's contraction
#
#   python ${ORBX_ROOT}/tools/make_validate_set.py validate_set.bin
#   ./validate_opencv --set validate_set.bin            # primitives only, no GPU needed (drop "orbx" from the link line then)
#   ./validate_opencv --set validate_set.bin --orbx     # + whole operator() on the MI355X
#
# Without CMake:
#   g++ -O3 -march=native -std=c++14 -DORBX_VALIDATE_REFERENCE -DORBX_VALIDATE_EXTRAS -I$ORB_SLAM3/include -I$ORBX_ROOT/include \
#       $(pkg-config --cflags opencv4) $ORBX_ROOT/tools/validate_opencv.cpp $ORB_SLAM3/src/ORBextractor.cc -o validate_opencv \
#       $(pkg-config --libs opencv4) -L$ORBX_ROOT/oracle -lorb_oracle -L$ORBX_ROOT/orb_slam3_modified_amd -lorbx \
#       -Wl,-rpath,$ORBX_ROOT/oracle -Wl,-rpath,$ORBX_ROOT/orb_slam3_modified_amd
