// TEST INFRASTRUCTURE: stand-in for the Boost.Serialization names DBoW2's headers use (BowVector.h:17-18,62-67,
// FeatureVector.h:18-19,27-32): `access` forwards to the class's private serialize() exactly like Boost's does, `base_object`
// returns the base sub-object.  Enough for a test archive (tests/support/serialize_check.cpp) to run `ar & bow; ar & fv;` the way
// include/KeyFrame.h:130-131 does; not a serialization library.
#pragma once
namespace boost { namespace serialization {
class access {
 public:
  template <class Archive, class T>
  static void serialize(Archive& ar, T& t, const unsigned int version) { t.serialize(ar, (int)version); }
};
template <class Base, class Derived> Base& base_object(Derived& d) { return static_cast<Base&>(d); }
} }
