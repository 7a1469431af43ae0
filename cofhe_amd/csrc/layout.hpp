// layout.hpp -- device-resident tensor layout shared by the kernels and the host packer.
//
// A form is one fixed-size record of 168 little-endian u32 words (672 B, 16-byte aligned):
//   words [  0,  40)  a      (magnitude, 1280-bit capacity)
//   words [ 40,  80)  |b|
//   words [ 80, 160)  c      (2560-bit capacity)
//   word   160        1 when b < 0
//   words [161, 168)  zero
// A ciphertext is two consecutive records (c1, c2); a tensor of E ciphertexts is 2E records
// in row-major element order, so the 8 limb groups of a wavefront stream 8 consecutive
// records (5.25 KiB contiguous) and lane gl of a group owns words [5gl, 5gl+5) of every
// 40-word plane -- the register layout of mp.hpp, no shuffles on load.
#pragma once
#include <stdint.h>
namespace cofhe {
constexpr int REC_WORDS = 168;
constexpr int REC_A = 0, REC_B = 40, REC_C = 80, REC_SIGN = 160;
}  // namespace cofhe
