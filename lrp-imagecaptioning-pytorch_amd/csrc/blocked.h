// The BLOCKED layout of the relevance chain's tensors in mode 3 (S between the layers, the per-image multiplicands xz):
//
//     [16-channel chunk][block of 32 consecutive pixels][part k = 0..3][pixel in block][4 channels]        (fp32)
//
// element (pixel gp, channel c) of a tensor of NP pixels sits at float offset
//     (c / 16) * CS + (gp / 32) * 512 + ((c % 16) / 4) * 128 + (gp % 32) * 4 + (c % 4),     CS = ceil(NP / 32) * 512.
// Why: in the transposed MFMA result (weights as the A operand) a lane owns ONE pixel and 16 channels - one 16-channel slice,
// with the weight rows permuted at pack time - and the consumer's staging item is one pixel's 16-channel slice too.  In NHWC a
// lane's four 16-byte accesses of such a slice land in four different 128-byte lines and a wave instruction touches 32 lines;
// here instruction k of 32 lanes covers 512 contiguous bytes: 8 full lines per wave instruction, as the LDS-transposed float4
// epilogue, without the LDS round trip.  S tensors number their pixels over all maps (gp = map * P + p), the
// multiplicands per image (each image its own block set: maps of one image share them).
#pragma once

namespace lrpx {

__host__ __device__ __forceinline__ long blk_chunk_stride(long n_pix) { return ((n_pix + 31) >> 5) * 512; }   // floats
// float offset of (pixel gp, channel quad 0 of chunk 0); + k * 128 for part k, + chunk * CS
__host__ __device__ __forceinline__ long blk_pix_off(long gp) { return ((gp >> 5) << 9) + ((gp & 31) << 2); }
__host__ __device__ __forceinline__ unsigned blk_pix_off32(int gp) { return ((unsigned)(gp >> 5) << 9) + ((unsigned)(gp & 31) << 2); }   // < 2^27 pixels
__host__ __device__ __forceinline__ long blk_off(long gp, int c, long cs) {
    return (long)(c >> 4) * cs + blk_pix_off(gp) + (((c & 15) >> 2) << 7) + (c & 3);
}
// floats of a blocked tensor of n_pix pixels x c channels (c % 16 == 0)
__host__ __device__ __forceinline__ long blk_floats(long n_pix, int c) { return (long)(c / 16) * blk_chunk_stride(n_pix); }

}  // namespace lrpx
