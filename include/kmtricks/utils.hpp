// kmtricks/utils.hpp -- the part of the kmtricks public utilities a merge plugin sees
// (reference include/kmtricks/utils.hpp:311-327): the count type selected by DMAX_C.
// Shipped so that existing plugin sources compile unchanged against the kmx driver.
#pragma once
#include <cstddef>
#include <cstdint>

namespace km {

template <size_t C> struct requiredC { enum { value = C <= 0xFF ? 8 : C <= 0xFFFF ? 16 : 32 }; };
template <int bits> struct select_;
template <> struct select_<8> { typedef uint8_t type; };
template <> struct select_<16> { typedef uint16_t type; };
template <> struct select_<32> { typedef uint32_t type; };
template <size_t C> struct selectC : select_<requiredC<C>::value> {};

}  // namespace km
