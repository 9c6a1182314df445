// fp32 -> three bf16 terms (x0 + x1 + x2 == x exactly, round to nearest each) and the "k16 image"
// of an fp32 matrix that the image GEMMs of gemm3.hip consume:
//     image[row / 32][k / 16][row % 32][plane 0..2][k % 16]  bf16
// i.e. 96 bytes per row and 16-deep K step, the 32 rows of a block next to each other: one (row block,
// K step) chunk = 3072 contiguous bytes = what three LDS-DMA instructions move, and consecutive K steps
// of a row block follow each other (a tile's operand stream is sequential in HBM / L2: the row-major
// form [row][k / 16][..] put every row of a step on the same few L2 channels - measured 1.6x slower DMA).
// K is zero-padded to a whole step, rows to a whole block (padding rows are never read as valid data).
// A producer that owns 8 consecutive k of a row writes three 16-byte pieces; one that owns 16 writes 96
// contiguous bytes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace marl {

constexpr int kImgStep = 16;       // K depth of one image step
constexpr int kImgRowBytes = 96;   // bytes of one row in one step: 3 planes x 16 bf16

typedef __bf16 split_bf16x2 __attribute__((ext_vector_type(2)));
typedef float split_f32x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> one dword (x low half, y high half) of bf16, round to nearest even
__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    const split_f32x2 v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, split_bf16x2));  // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    p2 = pack_bf16(rx - __uint_as_float(p1 << 16), ry - __uint_as_float(p1 & 0xffff0000u));
}

constexpr int kImgBlockRows = 32;
constexpr int kImgChunkBytes = kImgBlockRows * kImgRowBytes;  // 3072: one (row block, K step)
inline int img_steps(int k) { return (k + kImgStep - 1) / kImgStep; }
inline size_t img_bytes(int64_t rows, int k) {
    return (size_t)((rows + kImgBlockRows - 1) / kImgBlockRows) * img_steps(k) * kImgChunkBytes;
}
// byte offset of (row, K step) inside an image of `steps` steps
__host__ __device__ __forceinline__ size_t img_off(int64_t row, int step, int steps) {
    return ((size_t)(row >> 5) * steps + step) * kImgChunkBytes + (size_t)(row & 31) * kImgRowBytes;
}

// eight consecutive k (k0 % 8 == 0) of one row -> the three planes; step_img = image + img_off(row, k0 / 16, steps)
__device__ __forceinline__ void img_store8(char* step_img, int k0, const float (&v)[8]) {
    uint32_t p[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], p[0][i], p[1][i], p[2][i]);
    char* d = step_img + (k0 & 8) * 2;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<uint4*>(d + pl * 32) = make_uint4(p[pl][0], p[pl][1], p[pl][2], p[pl][3]);
}
// sixteen consecutive k (one whole step) of one row: 96 contiguous bytes
__device__ __forceinline__ void img_store16(char* step_img, const float (&v)[16]) {
    uint32_t p[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_pair(v[2 * i], v[2 * i + 1], p[0][i], p[1][i], p[2][i]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        *reinterpret_cast<uint4*>(step_img + pl * 32) = make_uint4(p[pl][0], p[pl][1], p[pl][2], p[pl][3]);
        *reinterpret_cast<uint4*>(step_img + pl * 32 + 16) = make_uint4(p[pl][4], p[pl][5], p[pl][6], p[pl][7]);
    }
}
// four consecutive k (k0 % 4 == 0)
__device__ __forceinline__ void img_store4(char* step_img, int k0, float a, float b, float c, float d4) {
    uint32_t a0, a1, a2, b0, b1, b2;
    split_pair(a, b, a0, a1, a2);
    split_pair(c, d4, b0, b1, b2);
    char* d = step_img + (k0 & 12) * 2;
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(d + 32) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + 64) = make_uint2(a2, b2);
}

}  // namespace marl
