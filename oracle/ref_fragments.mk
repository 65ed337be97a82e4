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
#   _ref/libref_orbextractor.so src/ORBextractor.cc + include/ORBextractor.h, unmodified, against the container shim; the five
#                               OpenCV algorithms it calls forward to the oracle's isolated primitives (liborb_oracle.so)
REFROOT ?= /root/reference
REF ?= $(REFROOT)/Thirdparty/DBoW2
CXX ?= g++
CXXFLAGS ?= -O2 -std=c++14 -fPIC -ffp-contract=off -w
SRCS := $(REF)/DBoW2/BowVector.cpp $(REF)/DBoW2/FeatureVector.cpp $(REF)/DBoW2/ScoringObject.cpp $(REF)/DBoW2/FORB.cpp \
        $(REF)/DUtils/Random.cpp $(REF)/DUtils/Timestamp.cpp ref_wrap.cpp
WORLD := ../tests/support/ref_world
WORLD_HDRS := $(wildcard $(WORLD)/*.h $(WORLD)/*/* $(WORLD)/*/*/*/*) $(wildcard ref_shims/opencv2/*/*.hpp)

all: _ref/libref_dbow2.so _ref/ref_matcher_world _ref/libref_orbextractor.so _ref/ref_streamed_frontend _ref/ref_kfdb_world

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
.PHONY: all
