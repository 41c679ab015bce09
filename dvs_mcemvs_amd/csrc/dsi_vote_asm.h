// dsi_vote_asm.h -- the two hand-scheduled gfx950 pieces every banded voting stream is built from: the GATHER of a batch
// (64 lanes: a 12-byte record, the 16 + 4 bytes of its packet's plane coefficients) and the VOTE of a batch (the transfer
// of mapper_emvs_stereo.cpp:194-195 with the residual-corrected divide, the accept test of cartesian3dgrid.h:255-259 on the
// band's rows, the four Q.31 bilinear weights of :261-270 and four ds_add_u64).  Shared by dsi_kernels.hip (the product
// streams, which differ in how they decide WHICH record a lane takes) and tools/vote_ceiling_bench.hip (the ceiling
// replica: the same two pieces with no bookkeeping at all).
//
// Operand contract of the enclosing asm statement:
//   %0 sxy (records)  %1 coef4 (plane coefficients, 32 B per packet)  %9 nx * 8  %10 LDS byte address of voxel (0, 0) of
//   the band's row 0  %11 nx - 2  %12 Li (first accepted row)  %13 Ui - 1 - Li (accepted rows - 1)
//   v40 = record index per lane (packet = index >> 10);  v36-v39, v58-v63 temporaries;  s50 is set by GATHER.
#pragma once

#define DSI_ASM_GATHER(EV, CA, CR)                                                                 \
    "v_mul_lo_u32 v58, v40, 12\n\t"       /* byte offset of the record */                           \
    "v_lshrrev_b32 v59, 5, v40\n\t"                                                                 \
    "v_and_b32 v59, 0x7ffffe0, v59\n\t"   /* byte offset of the packet's coefficients */            \
    "global_load_dwordx3 " EV ", v58, %0\n\t"                                                       \
    "global_load_dwordx4 " CA ", v59, %1\n\t"                                                       \
    "global_load_dword " CR ", v59, %1 offset:16\n\t"                                               \
    "s_mov_b32 s50, 1\n\t"              /* (three loads behind the cut-word prefetch) */

#define DSI_ASM_VOTE(EX, EY, EM, KA, KBX, KBY, KD, KR)                                             \
    "v_mul_f32 v58, " EX ", " KA "\n\t"                                                             \
    "v_mul_f32 v59, " EY ", " KA "\n\t"                                                             \
    "v_add_f32 v58, v58, " KBX "\n\t"     /* x0*a + bx */                                           \
    "v_add_f32 v59, v59, " KBY "\n\t"     /* y0*a + by */                                           \
    "v_mul_f32 v60, v58, " KR "\n\t"      /* div_rc: q = n*r */                                     \
    "v_mul_f32 v61, v59, " KR "\n\t"                                                                \
    "v_fma_f32 v62, -" KD ", v60, v58\n\t"                                                          \
    "v_fma_f32 v63, -" KD ", v61, v59\n\t"                                                          \
    "v_fma_f32 v60, v62, " KR ", v60\n\t"                                                           \
    "v_fma_f32 v61, v63, " KR ", v61\n\t"                                                           \
    "v_fma_f32 v62, -" KD ", v60, v58\n\t"                                                          \
    "v_fma_f32 v63, -" KD ", v61, v59\n\t"                                                          \
    "v_fma_f32 v60, v62, " KR ", v60\n\t" /* X */                                                   \
    "v_fma_f32 v61, v63, " KR ", v61\n\t" /* Y */                                                   \
    "v_cvt_flr_i32_f32 v58, v60\n\t"      /* xi */                                                  \
    "v_cvt_flr_i32_f32 v59, v61\n\t"      /* yi */                                                  \
    "v_subrev_u32 v63, %12, v59\n\t"      /* yi-Li */                                               \
    "v_cmpx_ge_u32 vcc, %11, v58\n\t"     /* exec &= 0 <= xi <= nx-2       (unsigned compare) */    \
    "v_cmpx_ge_u32 vcc, %13, v63\n\t"     /* exec &= 0 <= yi-Li <= Ui-1-Li (unsigned compare) */    \
    "v_fract_f32 v60, v60\n\t"            /* fx (X >= 0 here) */                                    \
    "v_fract_f32 v61, v61\n\t"            /* fy */                                                  \
    "v_lshl_add_u32 v58, v58, 3, %10\n\t"                                                           \
    "v_mad_i32_i24 v59, v59, %9, v58\n\t" /* LDS byte address of voxel (xi, yi) */                  \
    "v_sub_f32 v63, 1.0, v61\n\t"         /* 1-fy */                                                \
    "v_mul_f32 v60, 0x4f000000, v60\n\t"  /* fx * 2^31 */                                           \
    "v_sub_f32 v62, 0x4f000000, v60\n\t"  /* 2^31 - fx*2^31 == fl(1-fx) * 2^31 (power-of-two scale) */ \
    "v_mul_f32 v36, v62, v63\n\t"                                                                   \
    "v_mul_f32 v37, v60, v63\n\t"                                                                   \
    "v_mul_f32 v38, v62, v61\n\t"                                                                   \
    "v_mul_f32 v39, v60, v61\n\t"                                                                   \
    "v_cvt_u32_f32 v36, v36\n\t"                                                                    \
    "v_mad_u64_u32 v[62:63], vcc, v36, " EM ", 0\n\t"                                               \
    "ds_add_u64 v59, v[62:63]\n\t"                                                                  \
    "v_cvt_u32_f32 v37, v37\n\t"                                                                    \
    "v_mad_u64_u32 v[60:61], vcc, v37, " EM ", 0\n\t"                                               \
    "ds_add_u64 v59, v[60:61] offset:8\n\t"                                                         \
    "v_add_u32 v58, %9, v59\n\t"          /* next row */                                            \
    "v_cvt_u32_f32 v38, v38\n\t"                                                                    \
    "v_mad_u64_u32 v[62:63], vcc, v38, " EM ", 0\n\t"                                               \
    "ds_add_u64 v58, v[62:63]\n\t"                                                                  \
    "v_cvt_u32_f32 v39, v39\n\t"                                                                    \
    "v_mad_u64_u32 v[60:61], vcc, v39, " EM ", 0\n\t"                                               \
    "ds_add_u64 v58, v[60:61] offset:8\n\t"                                                         \
    "s_mov_b64 exec, -1\n\t"

// Lane mapping 8 (round 6, opt-in): the same vote into PAIRED 32-bit cells -- ONE ds_add_u64 updates the two cells (x, x + 1)
// of a row, so a vote is two LDS atomics instead of four.  A row of the band holds rw = (nx >> 1) + 1 words W_k = (cell 2k |
// cell 2k + 1 << 32) followed by rw words V_k = (cell 2k + 1 | cell 2k + 2 << 32): a vote with even xi goes to W[xi / 2], with
// odd xi to V[xi >> 1]; a cell's sum is its W part + its V part (flush_band_paired).  A record of multiplicity m adds
// round(m * w * 2^19) per cell (v_cvt_rpi_i32_f32 = floor(x + 0.5); the scale S = m * 2^19 is exact in fp32, the products carry
// fp32's 2^-24 relative rounding; < 2^30: never carries into the neighbour field);
// a sub-cell holds 8,192 full votes, the flush flags any sub-cell at or above 2^31 (4,096).  NOT the exact Q33.31 sums
// of the other mappings: accuracy ~ the reference's own fp32 accumulation (cartesian3dgrid.h:261-270).
//   %9 = bytes per band row (16 * rw), %10 = LDS byte address of word 0 of the band's row 0 - row_base rows,
//   %21 = 8 * rw (byte offset of the V words in a row); the rest as DSI_ASM_VOTE.
#define DSI_ASM_VOTE_PAIRED19(EX, EY, EM, KA, KBX, KBY, KD, KR)                                    \
    "v_mul_f32 v58, " EX ", " KA "\n\t"                                                             \
    "v_mul_f32 v59, " EY ", " KA "\n\t"                                                             \
    "v_add_f32 v58, v58, " KBX "\n\t"                                                               \
    "v_add_f32 v59, v59, " KBY "\n\t"                                                               \
    "v_mul_f32 v60, v58, " KR "\n\t"                                                                \
    "v_mul_f32 v61, v59, " KR "\n\t"                                                                \
    "v_fma_f32 v62, -" KD ", v60, v58\n\t"                                                          \
    "v_fma_f32 v63, -" KD ", v61, v59\n\t"                                                          \
    "v_fma_f32 v60, v62, " KR ", v60\n\t"                                                           \
    "v_fma_f32 v61, v63, " KR ", v61\n\t"                                                           \
    "v_fma_f32 v62, -" KD ", v60, v58\n\t"                                                          \
    "v_fma_f32 v63, -" KD ", v61, v59\n\t"                                                          \
    "v_fma_f32 v60, v62, " KR ", v60\n\t" /* X */                                                   \
    "v_fma_f32 v61, v63, " KR ", v61\n\t" /* Y */                                                   \
    "v_cvt_flr_i32_f32 v58, v60\n\t"      /* xi */                                                  \
    "v_cvt_flr_i32_f32 v59, v61\n\t"      /* yi */                                                  \
    "v_subrev_u32 v63, %12, v59\n\t"                                                                \
    "v_cmpx_ge_u32 vcc, %11, v58\n\t"                                                               \
    "v_cmpx_ge_u32 vcc, %13, v63\n\t"                                                               \
    "v_fract_f32 v60, v60\n\t"            /* fx */                                                  \
    "v_fract_f32 v61, v61\n\t"            /* fy */                                                  \
    "v_and_b32 v62, 1, v58\n\t"           /* odd xi: the V words */                                 \
    "v_lshrrev_b32 v58, 1, v58\n\t"                                                                 \
    "v_lshl_add_u32 v58, v58, 3, %10\n\t"                                                           \
    "v_mad_u32_u24 v58, v62, %21, v58\n\t"                                                          \
    "v_mad_i32_i24 v59, v59, %9, v58\n\t" /* LDS byte address of the word (xi, xi + 1) of row yi */ \
    "v_cvt_f32_u32 v62, " EM "\n\t"       /* the multiplicity goes into the scale: m <= 1024, exact */ \
    "v_sub_f32 v63, 1.0, v61\n\t"         /* 1-fy */                                                \
    "v_mul_f32 v62, 0x49000000, v62\n\t"  /* S = m * 2^19 (exact) */                                \
    "v_mul_f32 v60, v62, v60\n\t"         /* fx * S */                                              \
    "v_sub_f32 v62, v62, v60\n\t"         /* S - fx*S */                                            \
    "v_mul_f32 v36, v62, v63\n\t"                                                                   \
    "v_mul_f32 v37, v60, v63\n\t"                                                                   \
    "v_mul_f32 v38, v62, v61\n\t"                                                                   \
    "v_mul_f32 v39, v60, v61\n\t"                                                                   \
    "v_cvt_rpi_i32_f32 v36, v36\n\t"      /* m * w * 2^19 < 2^29: the field never carries */        \
    "v_cvt_rpi_i32_f32 v37, v37\n\t"                                                                \
    "ds_add_u64 v59, v[36:37]\n\t"                                                                  \
    "v_add_u32 v58, %9, v59\n\t"          /* next row */                                            \
    "v_cvt_rpi_i32_f32 v38, v38\n\t"                                                                \
    "v_cvt_rpi_i32_f32 v39, v39\n\t"                                                                \
    "ds_add_u64 v58, v[38:39]\n\t"                                                                  \
    "s_mov_b64 exec, -1\n\t"
