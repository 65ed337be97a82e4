// TEST INFRASTRUCTURE: stand-in for the Boost.Serialization declarations DBoW2's headers name
// (BowVector.h:17-18,52-57, FeatureVector.h:18-19,27-32); the serialize() templates are never instantiated.
#pragma once
namespace boost { namespace serialization {
class access;
template <class Base, class Derived> Base& base_object(Derived& d) { return static_cast<Base&>(d); }
} }
