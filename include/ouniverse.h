/*
 * ouniverse.h -- C ABI of libouniverse.so: MI355X (gfx950) implementation of line/open-universe's
 * `model.enhance` hot path (UNIVERSE / UNIVERSE++ reverse-diffusion sampler, score network, conditioner).
 *
 * The reference is pure Python/PyTorch and has no native interface; these entry points are what an FFI for
 * this path binds.  Each one names the reference interface it replaces (paths relative to the reference
 * checkout, tag 2024_10_08).  INTEGRATION.md shows the reference-side binding (ctypes).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / hip types (hipStream_t is passed as void*).
 *   - all tensors are fp32, contiguous, (B, C, T) with time innermost, resident on `device`.
 *   - the caller owns every input / output / workspace / packed-weight buffer; the library owns only
 *     the plan.  No allocation and no host synchronisation inside ou_condition / ou_score / ou_enhance:
 *     everything is enqueued on the caller's stream (hipGraph-capturable).
 *   - every function returns OU_OK (0) or a negative OU_E* code; ou_last_error() gives the message.
 *     Nothing throws across the ABI.
 *   - a handle is not re-entrant: one (process, device, stream) at a time.
 */
#ifndef OUNIVERSE_H
#define OUNIVERSE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OU_ABI_VERSION 5 /* 5: ou_enhance_var (batches whose rows have lengths of their own), workspace header carries the per-row geometry; ou_set_option / ou_get_option replace every OU_* environment switch (the library reads no environment variable), ou_config.fir_fold; 4: the packed blob carries a bf16-split weight copy (conv_split_kernel); 3: a Winograd-domain copy (round 5), ou_set_lane_batch, ou_lane_capacity */

enum {
  OU_OK = 0,
  OU_EINVAL = -1,    /* bad argument (Python side raises ValueError) */
  OU_ENOTIMPL = -2,  /* unsupported configuration (NotImplementedError) */
  OU_EMISSING = -3,  /* tensor missing from checkpoint (KeyError) */
  OU_ESHAPE = -4,    /* tensor shape mismatch */
  OU_EHIP = -5,      /* HIP runtime error */
  OU_ENOMEM = -6,    /* workspace too small */
  OU_ESYNC = -7      /* device-side timeout flag raised (GRU cluster exchange) */
};

enum { OU_KIND_UNIVERSE = 0, OU_KIND_UNIVERSE_GAN = 1 };
enum { OU_ACT_NONE = 0, OU_ACT_PRELU = 1, OU_ACT_SNAKE = 2 };

#define OU_MAX_RATES 8

/* One network's hyper-parameters == the constructor arguments of
 * ScoreNetwork (networks/universe/score.py:214-232) / ConditionerNetwork (condition.py:274-293). */
typedef struct ou_net_config {
  int32_t n_rates;
  int32_t rate_factors[OU_MAX_RATES];
  int32_t n_channels;
  int32_t fb_kernel_size;
  int32_t n_rff;
  int32_t noise_cond_dim;
  int32_t extra_conv_block;
  int32_t use_weight_norm;
  int32_t use_antialiasing;
  int32_t time_embedding_simple; /* 1: SimpleTimeEmbedding (sigma_block.py:60-78), 0: SigmaBlock RFF (:36-57) */
  int32_t n_mels;                /* conditioner only */
  int32_t n_mel_oversample;      /* conditioner only */
  int32_t encoder_gru_residual;  /* conditioner only */
} ou_net_config;

/* == config.model of the reference (the yaml files under config/model/), inference-relevant part;
 * replaces hydra `instantiate(config.model)` at inference_utils/model_loader.py:114. */
typedef struct ou_config {
  int32_t abi_version; /* OU_ABI_VERSION */
  int32_t kind;        /* OU_KIND_* : Universe (universe.py:44) / UniverseGAN (universe_gan.py:60) */
  int32_t fs;
  float level_db;      /* normalization_kwargs.level_db */
  int32_t has_edm;     /* edm: {noise: ...} present (universe.py:85-95) */
  float edm_noise;
  double sigma_min, sigma_max; /* diffusion.* (geometric schedule, universe.py:380-386) */
  int32_t use_signal_decoupling; /* universe_gan.py:117-126 */
  int32_t signal_decoupling_act; /* OU_ACT_* */
  ou_net_config score;
  ou_net_config cond;
  int32_t has_edm_data_level; /* edm.data_level_db given (universe.py:176-178); 0: sigma_data follows level_db */
  float edm_data_level_db;
  int32_t fir_fold; /* packing choice, not a reference hyper-parameter: bit 0 / bit 1 = fold the binomial anti-alias FIR of the
                     * down / up rate-change convs (blocks.py:213-227) into their weights (one launch less per rate change for 3x
                     * that conv's FLOPs; slower on MI355X, default 0).  The plan and the blob depend on it: same value for the
                     * packer and ou_create. */
  int32_t no_split_copy; /* packing choice: 1 = leave the bf16-split weight copy (conv_split_kernel: a quarter of the blob, PP16
                          * 162 of 648 MB) out -- for handles that never run a batch of 8 or more utterances per call (the
                          * launcher's rule selects that kernel from batch 8 / 16 on).  Same value for the packer and ou_create. */
} ou_config;

typedef struct ou_packer ou_packer;
typedef struct ou_handle ou_handle;
typedef void* ou_stream_t; /* hipStream_t */

const char* ou_version(void);
/* Message of the last error on this handle / packer (NULL handle: last global error). */
const char* ou_last_error(const ou_handle* h);
const char* ou_packer_last_error(const ou_packer* p);

/* ---- weights: replaces model.load_state_dict + EMA copy + (never called) remove_weight_norm ------------
 * model_loader.py:117-132, universe.py:841-865, blocks.py:36-50.
 * The packer takes tensors under the reference's state-dict keys (host fp32), folds weight-norm
 * (w = g*v/||v||), keeps the binomial anti-alias FIR of the rate-change convs (blocks.py:213-227) as separate taps, lays
 * every matrix out for the gfx950 kernels and returns one contiguous fp32 blob whose layout depends on the
 * config only.  The blob is what gets broadcast over RCCL and handed to ou_create(). */
int ou_packer_create(const ou_config* cfg, ou_packer** out);
int ou_packer_set(ou_packer* p, const char* key, const float* data, const int64_t* shape, int32_t ndim);
int ou_packer_finish(ou_packer* p, const float** blob_host, size_t* nbytes); /* blob owned by the packer */
void ou_packer_destroy(ou_packer* p);
int ou_packed_bytes(const ou_config* cfg, size_t* nbytes);

/* ---- model: replaces load_model()'s returned module (model_loader.py:62-137) ----------------------------
 * `weights_dev`: device pointer to the packed blob (caller-owned, must outlive the handle). */
int ou_create(const ou_config* cfg, const void* weights_dev, size_t nbytes, int32_t device, ou_handle** out);
void ou_destroy(ou_handle* h);

/* Workspace needed for a batch of B signals of padded length T (T % prod(rate_factors) == 0). */
int ou_workspace_bytes(const ou_handle* h, int32_t B, int32_t T, size_t* nbytes);
/* Once per workspace buffer, before its first use (and after every change of B): clears the header -- the sticky
 * device status word and the GRU exchange granules (whose tags continue from launch to launch, so the forward calls
 * themselves enqueue no memset).  Enqueued on `stream`.  The handle remembers (buffer, size, B): ou_condition,
 * ou_score and ou_enhance return OU_EINVAL for a workspace that was not prepared for their batch size.  A buffer
 * prepared for (B, T) serves every shorter length of the same batch size too (the header's layout depends on B alone):
 * a set of utterances of different lengths -- the reference CLI's loop over a directory, bin/enhance.py:173-192 -- runs on
 * ONE workspace sized for the longest; a buffer that is too small for a call is refused with OU_ENOMEM. */
int ou_workspace_init(ou_handle* h, int32_t B, int32_t T, void* ws, size_t ws_bytes, ou_stream_t stream);

/* Sampler constants, universe.py:301-311: sigma[n] (fp32, n = 0..n_steps-1), eta, beta. */
int ou_schedule(const ou_config* cfg, int32_t n_steps, double epsilon, float* sigma_out, double* eta, double* beta);

/* condition_model(mix, x_wav=mix, train=True)  -- universe.py:314-316 / condition.py:346-377.
 * `mix_norm`: (B,1,T) normalised mixture.  Results (cond[5], aux signal, latent) stay in the workspace
 * (see ou_tensor()) and are consumed by ou_score(). */
int ou_condition(ou_handle* h, const float* mix_norm, int32_t B, int32_t T, void* ws, size_t ws_bytes,
                 ou_stream_t stream);

/* score_model(x, sigma, cond) -> score  -- universe.py:286,197-209 (EDM wrapper) / score.py:277-297.
 * `sigma_host`: B floats on the host.  Requires a preceding ou_condition() on the same workspace. */
int ou_score(ou_handle* h, const float* x, const float* sigma_host, float* score_out, int32_t B, int32_t T,
             void* ws, size_t ws_bytes, ou_stream_t stream);

/* aux_to_wav(aux_signal) -- universe_gan.py:145-149 (alias-free Snake -> Conv1d(C0->1,k3)).
 * Reads the conditioner's aux signal from the workspace; writes (B,1,T). */
int ou_aux_to_wav(ou_handle* h, float* wav_out, int32_t B, int32_t T, void* ws, size_t ws_bytes,
                  ou_stream_t stream);

/* Flags of ou_enhance */
#define OU_ENH_KEEP_RMS 1u       /* universe.py:352-354 */
#define OU_ENH_USE_AUX_SIGNAL 2u /* universe.py:317-319 */
#define OU_ENH_NO_PEAK_GUARD 4u  /* skip universe.py:356-357 (debug) */
#define OU_ENH_SERIAL 8u         /* everything on the caller's stream, no side streams inside the call: the form to capture
                                  * into a hipGraph (a captured fork / join replays slower than the serial chain) */

/* Universe.enhance(mix, n_steps, epsilon, rng=...) -- universe.py:231-375, for a (B, T_raw) batch:
 * pad (:219-223) -> normalize (utils/norm.py:47-87) -> conditioner -> x0 = sigma_0 * noise[0] ->
 * N-1 x { score; x += sigma_n^2*eta*score + beta*sigma_{n+1}*noise[n+1] } -> last clean step ->
 * unpad -> [keep_rms] -> peak guard.  Ensemble replication / reduction stays with the caller.
 *   mix, out : (B, T_raw) device
 *   noise    : (n_steps - warm_start, B, T_pad) standard-normal, device, in the reference's draw order
 *              (x0, z_0 .. z_{N-2}); T_pad = T_raw + (tot_ds - T_raw % tot_ds)
 *   sigma_host: n_steps floats or NULL (then computed as ou_schedule does)
 *   warm_start: -1, or the step index to start from with x = aux_to_wav(aux) + noise (universe.py:328-331) */
int ou_enhance(ou_handle* h, const float* mix, float* out, const float* noise, int32_t B, int32_t T_raw,
               int32_t n_steps, double epsilon, const float* sigma_host, int32_t warm_start, uint32_t flags,
               void* ws, size_t ws_bytes, ou_stream_t stream);

/* The same for a batch whose rows have lengths of their OWN ("exact batching"; extension).  The reference has no such call:
 * its CLI runs a directory one file at a time (bin/enhance.py:173-192) and its collator zero-pads a batch WITHOUT a mask
 * (datasets/datamodule.py:24-42), so that the padding takes part in the normalisation, the mel norm, the conv halos and the GRU
 * -- every row then differs from what the file would give alone.  Here row b IS the call on that utterance alone, batched:
 * its own pad() split (universe.py:219-223: pad_b = tot_ds - t_raw[b] % tot_ds, pad_b / 2 in front), its own mean / std
 * (utils/norm.py:47-87) and mel norm (condition.py:105-106) over its own padded length, 'same' zero padding of every conv /
 * FIR right behind its own last sample on every level, GRU passes over its own frames (the backward pass starts at its own
 * last frame with h = 0), its own RMS restore and peak guard.  Results agree with the one-by-one loop to fp32 round-off
 * (the kernels a batch selects differ from the batch-1 ones; tests: >= 100 dB).
 *   mix, out : (B, T_raw_max) device; row b holds t_raw[b] samples, the rest of the row is ignored (mix) / zeroed (out)
 *   t_raw    : B lengths on the HOST, 1 <= t_raw[b] <= T_raw_max = max_b t_raw[b]
 *   noise    : (n_steps - warm_start, B, T_pad_max) device, T_pad_max = T_raw_max + (tot_ds - T_raw_max % tot_ds); row b
 *              uses its first t_raw[b] + pad_b columns (what a call on that row alone would draw), the rest is ignored
 *   workspace: as for ou_enhance with (B, T_pad_max).  A batch whose rows all have T_raw_max samples takes the plain path. */
int ou_enhance_var(ou_handle* h, const float* mix, float* out, const float* noise, int32_t B, int32_t T_raw_max,
                   const int32_t* t_raw, int32_t n_steps, double epsilon, const float* sigma_host, int32_t warm_start,
                   uint32_t flags, void* ws, size_t ws_bytes, ou_stream_t stream);

/* One sampler update on caller-owned buffers, for bindings that keep the reference's Python loop
 * (universe.py:339 `x = x + s_now^2 * eta * score + beta * z`, :343 `x = x + s_last^2 * score`):
 *   x[i] += c1 * score[i] + c2 * z[i]     over n = B*T elements; z may be NULL (last step).
 * Inside ou_enhance the same update is fused into the output conv of the score network. */
int ou_sampler_step(ou_handle* h, float* x, const float* score, const float* z, float c1, float c2, size_t n,
                    ou_stream_t stream);

/* ---- signal transform of the STFT-domain configurations: CompressedMagSTFT / CompressedMagSTFTPadded
 * (layers/dyn_range_comp.py:51-225; `model.transform`, universe.py:112-115, 274, 346).  Stateless: no handle.
 *   transform_type: 0 "none", 1 "exponent" ((1e-7 + |s|)^(e-1) s factor), 2 "log" (log(1 + |s|) sgn(s) factor)
 *   forward : x (B, T) -> out (B, 2F, n_frames), real parts then imaginary parts as channels, F = n_fft/2 + 1,
 *             STFT with center=True, zero padding, onesided, `window` (n_fft) on the device; n_frames = ou_transform_frames
 *   inverse : spec (B, 2F, n_frames) -> y (B, length): expansion, iSTFT (overlap-add / window envelope, torch.istft
 *             semantics, centre trimmed); scratch = B * n_frames * n_fft floats (device) */
int ou_transform_frames(int32_t T, int32_t n_fft, int32_t hop);
int ou_transform_forward(const float* x, int32_t B, int32_t T, const float* window, int32_t n_fft, int32_t hop,
                         int32_t transform_type, float abs_exponent, float factor, float* out, ou_stream_t stream);
int ou_transform_inverse(const float* spec, int32_t B, int32_t n_frames, const float* window, int32_t n_fft, int32_t hop,
                         int32_t transform_type, float abs_exponent, float factor, int32_t length, float* y,
                         float* scratch, ou_stream_t stream);

/* How the GRU clusters publish the hidden state (score.py:83-89,116 / condition.py:173-179,212 as a multi-workgroup
 * recurrence): 0 (default) = plain stores inside a cluster whose workgroups share one XCD (its L2 is their point of
 * coherence; verified by a rendezvous at every launch), agent-scope stores otherwise; 1 = agent-scope (sc1,
 * write-through) stores always (+0.1 ms per 401-frame pass).  Both forms are backed by a bounded-spin safety net that
 * repeats a publish system-scope and counts the event in the workspace header (word 20: every wait it cut short, late
 * members included; word 33: those where the publish was there for a system-scope load but not for the gather's
 * agent-scope load).  The library switches to 1 for good the first time word 33 moves (ou_check_device_status). */
int ou_set_gru_publish_mode(ou_handle* h, int32_t agent_scope);

/* ... and what the handle currently uses (1 after ou_set_gru_publish_mode(h, 1) or after ou_check_device_status found
 * that the safety net had to act).  A hipGraph captured earlier keeps the form it was captured with: re-capture. */
int ou_get_gru_publish_mode(const ou_handle* h);

/* K enhance calls in flight side by side in ONE process (extension; the reference's loop over files is serial,
 * bin/enhance.py:173-192): create K handles on the same weight blob, give each its own stream and workspace, and tell
 * every handle `lanes` = K and its own index `lane` (0 .. K - 1) BEFORE its first forward call.  A handle itself stays
 * non-re-entrant.  What the numbers are for: the workgroups of a GRU cluster wait for each other, so every GRU launch that
 * can be on the device at a time has to fit there whole -- the library sizes the launches of a lane to its share of the
 * XCDs and deals the clusters of lane l to XCDs of their own (2 B l, 2 B l + 1, ..).  Results do not depend on the lane:
 * a call keeps the kernels, tilings and -- as long as every cluster is still resident with it -- the split of the hidden
 * units of the single-lane call, i.e. bit-identical results (every shipped configuration at 1 .. 8 lanes; if the split does
 * not fit, the recurrence falls back to 16 units per workgroup: equal to fp32 rounding).  1 <= lanes <= 8. */
int ou_set_lanes(ou_handle* h, int32_t lanes, int32_t lane);

/* Lanes that run calls of DIFFERENT batch sizes at the same time (distributed.enhance_sharded(batch_size > 1, in_flight > 1)
 * on a ragged set, the CLI with files of different channel counts): the share of an XCD a lane's GRU launch may take and the
 * XCDs its clusters are dealt to were functions of that call's own B -- lanes that disagree about B then disagree about the
 * layout, and more cluster workgroups than an XCD holds wait for members that cannot be scheduled (a spurious device-side
 * time-out).  Tell EVERY handle of the pool the largest batch size any lane will run (`max_batch` >= 1; 0 = "this call's own
 * B", the default and the right value when all calls have one size): shares and placement are then computed from that one
 * number on all lanes.  A call smaller than max_batch may fall back to 16 hidden units per workgroup where the single call
 * takes 8 -- equal to fp32 rounding (> 100 dB), not bit-identical; calls of size max_batch are unchanged. */
int ou_set_lane_batch(ou_handle* h, int32_t max_batch);

/* How many lanes this device can carry when every lane may run a call of `max_batch` utterances: the GRU launches of all
 * lanes must be resident together (1 .. 8; e.g. UNIVERSE++ 16 kHz, H = 256: 8 lanes up to batch 2, 4 lanes at batch 4, 2 at
 * batch 8).  A pool larger than this makes the forward calls fail with OU_EHIP ("invalid configuration") -- the Python pool
 * (lanes.LanePool) clamps itself to this number. */
int ou_lane_capacity(const ou_handle* h, int32_t max_batch);

/* After the stream has been synchronised: OU_OK, or OU_ESYNC if a device-side timeout flag was raised.  The status
 * word (first 4 bytes of the workspace) is sticky: it stays raised until ou_workspace_init() / this call clears it.
 * Also reads status word 33 -- GRU publishes that were INVISIBLE to the gather's agent-scope loads (the recovery counter,
 * word 20, also counts cluster members that were merely late and does not switch anything): once word 33 has moved, the
 * handle publishes with agent-scope stores from the next call on (see ou_set_gru_publish_mode). */
int ou_check_device_status(ou_handle* h, void* ws);

/* ---- steering (tests, tuning) ---------------------------------------------------------------------------------
 * Typed options of a handle.  They replace the ~35 OU_* environment variables earlier ABI versions read: the library reads NO
 * environment variable any more, so nothing outside the caller's own code can change which kernels run.  Normal use needs
 * none of them -- the defaults are what the launchers' measured rules pick; the tests use them to force every kernel family
 * onto every layer it admits, the tools to sweep.  Keys: ou_option_name(0 .. ou_option_count() - 1), each with a one-line
 * ou_option_doc; integer-valued except `tile_min`.  Options that make a call return WRONG results by design (phase
 * ablation) exist in `make EXPERIMENTS=1` builds only (OU_ENOTIMPL otherwise).  A forward call works on the values it finds
 * when it starts; a captured hipGraph keeps the kernels it was captured with.  ou_plan_json echoes the current values. */
int ou_set_option(ou_handle* h, const char* key, double value); /* OU_EMISSING: no such key; OU_EINVAL: not an integer */
int ou_get_option(const ou_handle* h, const char* key, double* value);
int ou_reset_options(ou_handle* h); /* every option back to its default */
int ou_option_count(void);
const char* ou_option_name(int32_t index);
const char* ou_option_doc(int32_t index);
double ou_option_default(int32_t index);

/* ---- introspection (tests, profiling) ------------------------------------------------------------------- */
/* JSON description of the packed layers (name, kind, shapes, offsets into the blob); for a handle also "options": the current
 * value of every ou_set_option key. */
const char* ou_plan_json(const ou_handle* h);
const char* ou_packer_plan_json(const ou_packer* p);
/* Locate a named intermediate of the last ou_condition / ou_score / ou_enhance call in the workspace:
 * byte offset, channels, length (per batch element; layout (B, C, T)).  Names: see DESIGN.md. */
int ou_tensor(const ou_handle* h, const char* name, size_t* byte_offset, int32_t* C, int32_t* T);
/* Number of kernels the last forward enqueued, and the generic-conv launch count among them. */
int ou_launch_stats(const ou_handle* h, int32_t* n_launches, int32_t* n_conv_launches);
/* ---- input files of the CLI ------------------------------------------------------------------------------- */
/* FLAC stream decoder (pure host code).  Replaces the codec behind `torchaudio.load` in the reference's CLI
 * (open_universe/bin/enhance.py:183; AUDIO_SUFFIXES :33 lists .flac) -- torchaudio is not a dependency of this package.
 * Every checksum of the stream (frame-header CRC-8, frame CRC-16) is verified; the MD5 of the decoded audio is returned for the
 * caller to check (open_universe_amd/audio.py does).  ou_flac_last_error(): message of the last failure on this thread.
 *   ou_flac_info  : header fields; total_samples is counted by a decoding pass when the header does not carry it
 *   ou_flac_decode: out[channels][capacity_per_channel] <- the samples as integers (bits_per_sample wide, sign-extended) */
const char* ou_flac_last_error(void);
int ou_flac_info(const uint8_t* data, size_t bytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                 int64_t* total_samples, uint8_t* md5 /* 16 bytes, all zero = not recorded */);
int ou_flac_decode(const uint8_t* data, size_t bytes, int32_t* out, int64_t capacity_per_channel, int64_t* decoded);

/* Measurement / tuning entry points (ou_profile_*, ou_bench_conv) are declared in ouniverse_tuning.h. */

#ifdef __cplusplus
}
#endif
#endif /* OUNIVERSE_H */
