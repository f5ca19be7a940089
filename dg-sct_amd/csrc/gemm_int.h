// Internal seam between the two GEMM translation units (gemm.hip: the 4-wave tiled engine; gemm8.hip: the 8-wave
// LDS-DMA pipelined kernel for the deep products).  Not part of prims.h: plan.cpp only ever calls gemm().
#pragma once
#include "prims.h"

namespace dgsct {

// per-launch event timing of the GEMM family (bench.py's roofline leg); rec == nullptr when profiling is off
struct GemmProfShape { int M, N, K, KB, batch, splitk, cfg, ak, bk, atomic, wide; double bytes; };
void* gemm_prof_begin(void* stream, double flops, const GemmProfShape& shape);
void gemm_prof_end(void* rec, void* stream);

// Launches `g` on the 8-wave kernel when it qualifies (bf16, deep contraction, 256-row tiles, plain epilogue) and
// returns true; false = not eligible, the caller runs the tiled engine.
bool gemm8_try(const Ctx& ctx, const Gemm& g);
// 0: off, 1: on for shapes that fill the chip (default; DGSCT_GEMM8), 2: on for every eligible shape (tests).  set < 0: query.
int gemm8_mode(int set);

// Skinny products (M <= 256 rows, both operands K-major: the per-frame gate MLPs): one 32 x 32 tile per workgroup, contraction split
// over its four waves, operands straight to registers (gemm_skinny.hip).  Same contract as gemm8_try.
bool gemm_skinny_try(const Ctx& ctx, const Gemm& g);
int gemm_skinny_mode(int set);      // 0: off, 1: on (default; DGSCT_GEMM_SKINNY).  set < 0: query.

// Weight gradients over the token rows (both operands [rows][width], a small output, a very deep contraction): one stream over the
// two tensors, output tiles dealt to the waves (gemm_tall.hip).  Same contract as gemm8_try.  CONTRACT: MN-major operands handed to gemm() must point at
// COLUMN 0 of their [rows][ld] tensor (a column slab is selected with `bs`, not by offsetting the base pointer): this kernel reads
// whole ld-wide rows from the pointer, and an offset base would run past the end of the tensor in the last 64-row block.
bool gemm_tall_try(const Ctx& ctx, const Gemm& g);
int gemm_tall_mode(int set);        // 0: off, 1: on (default; DGSCT_NO_GEMM_TALL).  set < 0: query.

int gemm8_pipe_mode(int set);      // k-loop variant of gemm8 (1: pipelined across k-tiles, default; 0: two barriers per k-tile)
int gemm8_stag_mode(int set);      // (round 6) the two wave halves issue their LDS-DMA shares a quarter k-tile apart (1) or in lock-step (0)
int gemm8_wg_target(int set);      // experiment: workgroup target of late-stage weight gradients on gemm8 (0 = off)
int gemm_cfgx_mode(int set);       // tile-configuration experiments of the tiled engine (bit mask)
int gemm_noatomic_mode(int set);   // what-if (timing only): split-K partials as plain stores

}  // namespace dgsct
