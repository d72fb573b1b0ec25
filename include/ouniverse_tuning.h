/*
 * ouniverse_tuning.h -- measurement and tuning entry points of libouniverse.so.  NOT part of the drop-in boundary
 * (include/ouniverse.h): nothing here replaces a reference interface; bench.py and tools/ use them to time kernels.
 */
#ifndef OUNIVERSE_TUNING_H
#define OUNIVERSE_TUNING_H

#include "ouniverse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement: when enabled, every launch of the conv kernels (generic and fused) in subsequent forward calls records its own
 * duration on the device (first block start .. last block end, constant 100 MHz clock -- HIP events around single
 * launches also count the command-processor gaps and over-read by ~4 us).  ou_profile_read() synchronises the
 * device and returns, per launch, the ms, the layer's algorithmic FLOPs / bytes (reference, un-folded
 * accounting) and the tile config (>= 100: fused ConvBlock variants). */
int ou_profile_enable(ou_handle* h, int32_t on);
/* Tuning aid: time ONE packed conv layer (by its reference state-dict prefix) on synthetic data, optionally forcing
 * the tile configuration / chunks-per-stage; ms per launch from HIP events. */
int ou_bench_conv(ou_handle* h, const char* layer, int32_t B, int32_t Tin, int32_t cfg, int32_t sc, int32_t with_res,
                  int32_t iters, void* ws, size_t ws_bytes, ou_stream_t stream, float* ms_per_iter, int32_t* cfg_used);
int ou_profile_read(ou_handle* h, int32_t max_records, float* ms, double* flops, double* bytes, int32_t* cfg,
                    int32_t* n_records);
/* The raw stamps of the same records (first block start / last block end, 10 ns ticks of the device's constant clock -- one
 * clock for every handle of the process: the launches of several lanes lie on one timeline) and the variant code
 * (>= 1000: one GRU pass of cfg - 1000 steps). */
int ou_profile_read_ticks(ou_handle* h, int32_t max_records, uint64_t* t_start, uint64_t* t_end, int32_t* cfg,
                          int32_t* n_records);

/* Per-wave phase stamps of the fused ConvBlock kernel for ONE block of the network (its name as in ou_tensor, e.g.
 * "score.enc0"; NULL / "": off) -- written into the last 16 MB of the workspace (tools/chain_ts.py). */
int ou_set_stamp_layer(ou_handle* h, const char* block_name);

#ifdef __cplusplus
}
#endif
#endif /* OUNIVERSE_TUNING_H */
