# Compiles the parts of the REFERENCE that build from their own few source files (vendored DBoW2: vocabulary
# tree, BowVector, FeatureVector, scoring, FORB Hamming distance) straight from /root/reference, with shim
# headers for the two absent third-party packages (oracle/ref_shims: a minimal cv::Mat and a Boost.Serialization
# stub).  Output: oracle/_ref/libref_dbow2.so (git-ignored; travels to the GPU box).  Reference sources are
# never copied into this repository.  src/ORBextractor.cc and src/ORBmatcher.cc are NOT buildable this way
# (they call into OpenCV's algorithms / Eigen / Sophus; DESIGN.md "oracle").
REF ?= /root/reference/Thirdparty/DBoW2
CXX ?= g++
CXXFLAGS ?= -O2 -std=c++14 -fPIC -ffp-contract=off -w
SRCS := $(REF)/DBoW2/BowVector.cpp $(REF)/DBoW2/FeatureVector.cpp $(REF)/DBoW2/ScoringObject.cpp $(REF)/DBoW2/FORB.cpp \
        $(REF)/DUtils/Random.cpp $(REF)/DUtils/Timestamp.cpp ref_wrap.cpp

_ref/libref_dbow2.so: $(SRCS) ref_shims/opencv2/core/core.hpp ref_shims/boost/serialization/serialization.hpp
	mkdir -p _ref
	$(CXX) $(CXXFLAGS) -Iref_shims -I$(REF) -shared -o $@ $(SRCS)
