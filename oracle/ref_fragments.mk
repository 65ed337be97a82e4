# Compiles the parts of the REFERENCE that build from their own few source files straight from /root/reference, with
# shim headers for the absent third-party packages.  Outputs only into oracle/_ref/ (git-ignored; travels to the GPU box).
# Reference sources are never copied into this repository.
#   _ref/libref_dbow2.so        vendored DBoW2 (vocabulary tree, BowVector, FeatureVector, scoring, FORB distance) +
#                               oracle/ref_wrap.cpp; shims: oracle/ref_shims (container-only cv::Mat, Boost.Serialization stub)
#   _ref/ref_matcher_world      src/ORBmatcher.cc + include/ORBmatcher.h, unmodified, against the object model of
#                               tests/support/ref_world (Frame / KeyFrame / MapPoint / Eigen / Sophus stand-ins holding the
#                               members the matcher touches) + the scenario driver tests/support/matcher_world.cpp
#   _ref/ref_streamed_frontend  src/ORBextractor.cc + src/ORBmatcher.cc, unmodified, + the DBoW2 library above behind tools/streamed_frontend.cpp:
#                               the per-frame Tracking sequence on the reference's own CPU code (bench.py's streamed_frontend.cpu leg)
#   _ref/ref_kfdb_world         src/KeyFrameDatabase.cc + include/KeyFrameDatabase.h + include/ORBVocabulary.h (the DBoW2 template), unmodified,
#                               against the same object model (KeyFrame's database fields, covisibility accessors, a Map stand-in) + the
#                               scenario driver tests/support/kfdb_world.cpp; the reference's own BowVector.h / FeatureVector.h are
#                               force-included first so that ONE definition of the two DBoW2 classes is seen
#   _ref/ref_frame_world        src/Frame.cc + include/Frame.h, unmodified ("the drop-in boundary's only caller"), + src/ORBextractor.cc + the DBoW2 library
#                               above, against the stand-ins of tests/support/frame_world (IMU / Converter / Settings by include guard, cameras,
#                               MapPoint, a two-constant ORBmatcher) + the scenario driver tests/support/frame_world.cpp
#   _ref/dropin_frame_world     THE SAME src/Frame.cc over this repository's include/ORBextractor.h + include/ORBVocabulary.h, linked against
#                               orb_slam3_modified_amd/liborbx.so: what the GPU test runs (the GPU box has no /root/reference to compile from)
#   _ref/dropin_frame_world_cpu the same, linked against the oracle-backed C-ABI stub (tests/support/orbx_oracle_stub.cpp): the CPU test
#   _ref/dropin_frame_world_stereo[_cpu]  the two drop-in builds again with integration/Frame_stereo.patch applied to a TEMPORARY copy of src/Frame.cc
#                               (made by `patch -o` under $$TMPDIR, compiled, deleted: never stored) and -DORBX_DEVICE_STEREO: ComputeStereoMatches on the
#                               device pyramids, no host mirror of mvImagePyramid — the opt-in of INTEGRATION.md section 4, held to the same golden
#   _ref/libref_orbextractor.so src/ORBextractor.cc + include/ORBextractor.h, unmodified, against the container shim; the five
#                               OpenCV algorithms it calls forward to the oracle's isolated primitives (liborb_oracle.so)
#   _ref/libref_orbextractor_fma.so  the same, compiled with -O3 -mfma and the compiler's default FP contraction (a -march=native build)
#   _ref/validate_opencv        tools/validate_opencv.cpp + src/ORBextractor.cc over the shim + liborbx.so + the oracle (the harness of the
#                               maintainer's OpenCV check)
REFROOT ?= /root/reference
comma := ,
REF ?= $(REFROOT)/Thirdparty/DBoW2
CXX ?= g++
CXXFLAGS ?= -O2 -std=c++14 -fPIC -ffp-contract=off -w
SRCS := $(REF)/DBoW2/BowVector.cpp $(REF)/DBoW2/FeatureVector.cpp $(REF)/DBoW2/ScoringObject.cpp $(REF)/DBoW2/FORB.cpp \
        $(REF)/DUtils/Random.cpp $(REF)/DUtils/Timestamp.cpp ref_wrap.cpp
WORLD := ../tests/support/ref_world
WORLD_HDRS := $(wildcard $(WORLD)/*.h $(WORLD)/*/* $(WORLD)/*/*/*/*) $(wildcard ref_shims/opencv2/*/*.hpp)

all: _ref/libref_dbow2.so _ref/ref_matcher_world _ref/libref_orbextractor.so _ref/libref_orbextractor_fma.so _ref/ref_streamed_frontend _ref/ref_kfdb_world _ref/ref_frame_world \
     _ref/dropin_frame_world_cpu _ref/dropin_frame_world_stereo_cpu \
     $(if $(wildcard ../orb_slam3_modified_amd/liborbx.so),_ref/dropin_frame_world _ref/dropin_frame_world_stereo _ref/validate_opencv)

_ref/libref_dbow2.so: $(SRCS) ref_shims/opencv2/core/core.hpp ref_shims/boost/serialization/serialization.hpp
	mkdir -p _ref
	$(CXX) $(CXXFLAGS) -Iref_shims -I$(REF) -shared -o $@ $(SRCS)

# -include ref_world.h: defines the include guards of the reference's Frame.h / KeyFrame.h / MapPoint.h before its
# ORBmatcher.h includes them from its own directory, so the object model of tests/support/ref_world is the one seen
_ref/ref_matcher_world: $(REFROOT)/src/ORBmatcher.cc $(REFROOT)/include/ORBmatcher.h ../tests/support/matcher_world.cpp ../tests/support/world_scene.h $(WORLD_HDRS)
	mkdir -p _ref
	$(CXX) -O2 -std=c++17 -ffp-contract=off -w -include $(WORLD)/ref_world.h -I$(WORLD) -Iref_shims -I$(REFROOT)/include \
	    $(REFROOT)/src/ORBmatcher.cc ../tests/support/matcher_world.cpp -o $@

# the five OpenCV algorithm calls resolve to orbo_* in liborb_oracle.so (built by oracle/Makefile)
_ref/libref_orbextractor.so: $(REFROOT)/src/ORBextractor.cc $(REFROOT)/include/ORBextractor.h ref_wrap_extractor.cpp liborb_oracle.so $(wildcard ref_shims/opencv2/*/*.hpp)
	mkdir -p _ref
	$(CXX) -O2 -std=c++14 -fPIC -ffp-contract=off -w -Iref_shims -I$(REFROOT)/include -shared -o $@ $(REFROOT)/src/ORBextractor.cc ref_wrap_extractor.cpp \
	    -L. -lorb_oracle -Wl,-rpath,'$$ORIGIN/..'

# The same file the way the reference's CMakeLists.txt:10-13 builds it on an FMA machine: -O3 and -mfma (what -march=native adds that can change
# results) under the compiler's DEFAULT -ffp-contract — GCC / clang then contract the pattern rotation of src/ORBextractor.cc:118-120 into
# FMAs.  Pins the oracle's "brief_fma" variant (tests/test_opencv_variants.py); loaded only on hosts whose /proc/cpuinfo lists fma.
_ref/libref_orbextractor_fma.so: $(REFROOT)/src/ORBextractor.cc $(REFROOT)/include/ORBextractor.h ref_wrap_extractor.cpp liborb_oracle.so $(wildcard ref_shims/opencv2/*/*.hpp)
	mkdir -p _ref
	$(CXX) -O3 -mfma -std=c++14 -fPIC -w -Iref_shims -I$(REFROOT)/include -shared -o $@ $(REFROOT)/src/ORBextractor.cc ref_wrap_extractor.cpp \
	    -L. -lorb_oracle -Wl,-rpath,'$$ORIGIN/..'

# -O3 like the reference's own CMAKE_CXX_FLAGS_RELEASE (CMakeLists.txt:13-15; -march=native left out: the binary travels to the GPU box)
_ref/ref_streamed_frontend: $(REFROOT)/src/ORBextractor.cc $(REFROOT)/src/ORBmatcher.cc ../tools/streamed_frontend.cpp ../tests/support/world_scene.h \
                            $(WORLD_HDRS) _ref/libref_dbow2.so liborb_oracle.so
	mkdir -p _ref
	$(CXX) -O3 -std=c++17 -ffp-contract=off -w -include $(WORLD)/ref_world.h -I$(WORLD) -Iref_shims -I$(REFROOT)/include \
	    $(REFROOT)/src/ORBextractor.cc $(REFROOT)/src/ORBmatcher.cc ../tools/streamed_frontend.cpp -o $@ \
	    -L. -lorb_oracle -L_ref -lref_dbow2 -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,'$$ORIGIN'

_ref/ref_kfdb_world: $(REFROOT)/src/KeyFrameDatabase.cc $(REFROOT)/include/KeyFrameDatabase.h $(REFROOT)/include/ORBVocabulary.h \
                     ../tests/support/kfdb_world.cpp ../tests/support/world_scene.h $(WORLD_HDRS) _ref/libref_dbow2.so
	mkdir -p _ref
	$(CXX) -O2 -std=c++17 -ffp-contract=off -w -include $(REF)/DBoW2/BowVector.h -include $(REF)/DBoW2/FeatureVector.h -include $(WORLD)/ref_world.h \
	    -I$(WORLD) -Iref_shims -I$(REFROOT)/include -I$(REFROOT) $(REFROOT)/src/KeyFrameDatabase.cc ../tests/support/kfdb_world.cpp -o $@ \
	    -L_ref -lref_dbow2 -Wl,-rpath,'$$ORIGIN'

# -include prelude.h: defines the guards of include/ImuTypes.h, Converter.h, Settings.h (which include/Frame.h pulls from its own directory) and
# supplies the members src/Frame.cc uses; everything else is shadowed through the include path (tests/support/frame_world comes first)
FW := ../tests/support/frame_world
FW_HDRS := $(wildcard $(FW)/*.h $(FW)/*/* $(FW)/*/*/* $(FW)/*/*/*/*) $(WORLD)/Eigen/Core $(WORLD)/sophus/se3.hpp $(wildcard ref_shims/opencv2/*/*.hpp)
FW_FLAGS := -O3 -std=c++17 -ffp-contract=off -w -pthread -include $(FW)/prelude.h -I$(FW) -Iref_shims
_ref/ref_frame_world: $(REFROOT)/src/Frame.cc $(REFROOT)/include/Frame.h $(REFROOT)/src/ORBextractor.cc ../tests/support/frame_world.cpp $(FW_HDRS) \
                      _ref/libref_dbow2.so liborb_oracle.so
	mkdir -p _ref
	$(CXX) $(FW_FLAGS) -I$(REFROOT) -I$(REFROOT)/include -I$(REFROOT)/include/CameraModels \
	    $(REFROOT)/src/Frame.cc $(REFROOT)/src/ORBextractor.cc ../tests/support/frame_world.cpp -o $@ \
	    -L. -lorb_oracle -L_ref -lref_dbow2 -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,'$$ORIGIN'

# the drop-in builds: this repository's include/ first, so "ORBextractor.h" and (through the prelude, -DFRAME_WORLD_DROPIN) "ORBVocabulary.h"
# are the drop-in headers; src/Frame.cc and include/Frame.h are still the reference's files
# include/ORBVocabulary.h takes DBoW2::BowVector / FeatureVector from the reference's own two headers when they are on the include path
# (INTEGRATION.md section 3), so their two small source files are linked exactly as a maintainer's libDBoW2 would supply them
DROPIN_HDRS := ../include/ORBextractor.h ../include/ORBVocabulary.h ../include/orbx.h
DROPIN_DBOW2 := $(REF)/DBoW2/BowVector.cpp $(REF)/DBoW2/FeatureVector.cpp
_ref/dropin_frame_world_cpu: $(REFROOT)/src/Frame.cc $(REFROOT)/include/Frame.h ../tests/support/frame_world.cpp ../tests/support/orbx_oracle_stub.cpp \
                             $(FW_HDRS) $(DROPIN_HDRS) liborb_oracle.so
	mkdir -p _ref
	$(CXX) $(FW_FLAGS) -DFRAME_WORLD_DROPIN -I../include -I$(REFROOT) -I$(REFROOT)/include -I$(REFROOT)/include/CameraModels \
	    $(REFROOT)/src/Frame.cc $(DROPIN_DBOW2) ../tests/support/frame_world.cpp ../tests/support/orbx_oracle_stub.cpp -o $@ \
	    -L. -lorb_oracle -Wl,-rpath,'$$ORIGIN/..'

_ref/dropin_frame_world: $(REFROOT)/src/Frame.cc $(REFROOT)/include/Frame.h ../tests/support/frame_world.cpp $(FW_HDRS) $(DROPIN_HDRS) \
                         ../orb_slam3_modified_amd/liborbx.so liborb_oracle.so
	mkdir -p _ref
	$(CXX) $(FW_FLAGS) -DFRAME_WORLD_DROPIN -I../include -I$(REFROOT) -I$(REFROOT)/include -I$(REFROOT)/include/CameraModels \
	    $(REFROOT)/src/Frame.cc $(DROPIN_DBOW2) ../tests/support/frame_world.cpp -o $@ \
	    -L../orb_slam3_modified_amd -lorbx -L. -lorb_oracle -Wl,-rpath,'$$ORIGIN/../../orb_slam3_modified_amd' -Wl,-rpath,'$$ORIGIN/..'

# The opt-in: src/Frame.cc with integration/Frame_stereo.patch.  The patched text exists only as a temporary file next to nothing of ours; the
# reference's quoted includes ("Frame.h", "G2oTypes.h", ...) resolve through the include path as before.
STEREO_PATCH := ../integration/Frame_stereo.patch
define patched_frame_build
	mkdir -p _ref
	t=$$(mktemp -d) && patch -s -o $$t/Frame.cc $(REFROOT)/src/Frame.cc $(STEREO_PATCH) && \
	  { $(CXX) $(FW_FLAGS) -DFRAME_WORLD_DROPIN -DORBX_DEVICE_STEREO -I../include -I$(REFROOT) -I$(REFROOT)/include -I$(REFROOT)/include/CameraModels -I$(REFROOT)/src \
	    $$t/Frame.cc $(DROPIN_DBOW2) ../tests/support/frame_world.cpp $(1) -o $@ $(2); rc=$$?; rm -rf $$t; exit $$rc; }
endef
_ref/dropin_frame_world_stereo_cpu: $(REFROOT)/src/Frame.cc $(REFROOT)/include/Frame.h $(STEREO_PATCH) ../tests/support/frame_world.cpp ../tests/support/orbx_oracle_stub.cpp \
                                    $(FW_HDRS) $(DROPIN_HDRS) liborb_oracle.so
	$(call patched_frame_build,../tests/support/orbx_oracle_stub.cpp,-L. -lorb_oracle -Wl$(comma)-rpath$(comma)'$$ORIGIN/..')

_ref/dropin_frame_world_stereo: $(REFROOT)/src/Frame.cc $(REFROOT)/include/Frame.h $(STEREO_PATCH) ../tests/support/frame_world.cpp $(FW_HDRS) $(DROPIN_HDRS) \
                                ../orb_slam3_modified_amd/liborbx.so liborb_oracle.so
	$(call patched_frame_build,,-L../orb_slam3_modified_amd -lorbx -L. -lorb_oracle -Wl$(comma)-rpath$(comma)'$$ORIGIN/../../orb_slam3_modified_amd' -Wl$(comma)-rpath$(comma)'$$ORIGIN/..')
.PHONY: all

# tools/validate_opencv.cpp — the maintainer's OpenCV check (tools/validate_opencv.cmake) — built HERE over the container shim, with the
# reference's src/ORBextractor.cc compiled in for the operator() leg: proves the harness (the shim's cv:: functions are the oracle's), and on
# the GPU box compares the reference-compiled operator() with liborbx.so image by image (tests/test_validate_opencv.py)
_ref/validate_opencv: ../tools/validate_opencv.cpp $(REFROOT)/src/ORBextractor.cc $(REFROOT)/include/ORBextractor.h ../include/orbx.h ../include/orbx_cv_calibrate.h \
                      ../orb_slam3_modified_amd/liborbx.so liborb_oracle.so $(wildcard ref_shims/opencv2/*/*.hpp)
	mkdir -p _ref
	$(CXX) -O2 -std=c++14 -ffp-contract=off -w -DORBX_VALIDATE_REFERENCE -I$(REFROOT)/include -I../include -Iref_shims ../tools/validate_opencv.cpp $(REFROOT)/src/ORBextractor.cc -o $@ \
	    -L../orb_slam3_modified_amd -lorbx -L. -lorb_oracle -Wl,-rpath,'$$ORIGIN/../../orb_slam3_modified_amd' -Wl,-rpath,'$$ORIGIN/..' -Wl,--allow-shlib-undefined
