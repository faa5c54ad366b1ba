// Build-time knobs of the experiments recorded in DESIGN.md section 4.2 (tools/exp/*.sh build them as
//   make VARIANT=<name> EXTRA="-D..."  ->  sucre_amd/libsucre_hip_<name>.so,   selected at run time with SUCRE_HIP_LIB).
// The product build defines none of them; tests/test_host_logic.py compiles every one so that they cannot rot.
// They are variants of ONE code path (same kernels, same ABI), never alternative back ends.
#pragma once

// Ablations of the fit kernels: the arithmetic of a chunk compiled out (the data are still touched), the LDS-DMA copies
// compiled out (the ring keeps whatever LDS holds).  Results are meaningless; only the launch time is read.
#ifdef SUCRE_EXP_NOCOMPUTE
constexpr bool kExpNoCompute = true;
#else
constexpr bool kExpNoCompute = false;
#endif
#ifdef SUCRE_EXP_NOLOAD
constexpr bool kExpNoLoad = true;
#else
constexpr bool kExpNoLoad = false;
#endif
// J-parameter kernels: the stepped J and moments are computed and not written (how much of a launch is the write stream).
#ifdef SUCRE_EXP_NOSTORE
constexpr bool kExpNoStore = true;
#else
constexpr bool kExpNoStore = false;
#endif
// The stepped J (and moments) written with the streaming policy (nt): 0 never, 1 always, 2 in batch launches only (the product).
// Same box, round 5.  1080p x 65 views: 1 = 126.3-126.5 us alone against 127.2-127.4, 84.5-84.8 Mpix/s with two images in
// flight against 85.1-85.2 (nothing to gain: 8 % of a launch's bytes).  Config 1 in launches of 32 (35 % of the bytes are these
// stores): a launch alone 244.6-245.2 against 245.2-248.0 us, but 221.7-223.0 against 208.8-210.1 Mpix/s with two slots in
// flight (the other slot's match kernels find their views in L2); closed form 248.5-249.7 against 241.9-245.0.
#ifndef SUCRE_STORE_NT
#define SUCRE_STORE_NT 2
#endif
constexpr int kStoreNt = SUCRE_STORE_NT;

// Waves per SIMD the fit kernels are compiled for (= workgroups per CU of their persistent grids).  J-parameter kernel, same
// box, two rounds (round 5): 5 waves 126.6-127.1 us alone / 85.7 Mpix/s; 6 waves (77 registers; deal 64,52,42,32,24,16)
// 126.1-126.4 / 85.8-85.9; 6 waves with equal shares 130.4-130.7 / 85.4-85.6; round 4: 4 waves = 5 waves.  Not occupancy.
#ifndef SUCRE_FIT_WAVES
#define SUCRE_FIT_WAVES 5
#endif
#ifndef SUCRE_CLOSED_WAVES
#define SUCRE_CLOSED_WAVES 4
#endif

// Strips per fit wave of a small image (layout.h make_layout: the image's own grid is ceil(tiles / this), at most the persistent
// grid; 1 = the rule until round 5).
#ifndef SUCRE_MIN_STRIPS
#define SUCRE_MIN_STRIPS 4
#endif

// Slots of a wave's LDS-DMA ring (prefetch depth + 1) and the cache policy of its copies (" nt": items are read once per
// launch -- measured 20 % faster than the default policy; next to it " sc0", " sc1", " sc0 sc1": equal).
#ifndef SUCRE_RING
#define SUCRE_RING 3
#endif
#ifndef SUCRE_DMA_POLICY
#define SUCRE_DMA_POLICY " nt"
#endif

// match.hip: the pixel quotients x/z, y/z as two IEEE divisions everywhere (1) instead of x * rcp(z) with the divisions
// as the fallback near integers (0, the product; same match sets by construction -- DESIGN.md section 4.1; the A/B of
// tools/exp/ab_lib.sh exactdiv).
#ifndef SUCRE_EXACT_DIV
#define SUCRE_EXACT_DIV 0
#endif
constexpr bool kExactDiv = SUCRE_EXACT_DIV != 0;

// fit.hip / light.hip: Adam's step on a pixel's J with the IEEE divisions and square root of torch's formula (1) instead of
// the hardware reciprocal / square root (0, the product: fit_math.h adam_update_J; tools/exp/ab_bench.sh exactadam).
#ifndef SUCRE_EXACT_J_ADAM
#define SUCRE_EXACT_J_ADAM 0
#endif
constexpr bool kExactJAdam = SUCRE_EXACT_J_ADAM != 0;

// light.hip, --light-model --use-closed-form: I = k/255 folded into I - b of the J pass (1) / into the residual of the gradient
// pass (2), two instructions per channel each.  Round 6, one box: 479 -> 472 us per iteration each (both: 477 with the build of
// that hour), but the closed-form trajectories of the fixtures then leave the reference's: parameters 2.7e-4 away within 50
// iterations where the reference's own two runs differ by 1.7e-5 (tests/test_gpu_parity.py
// test_light_model_closed_form_vs_reference_golden, four tests red with either fold) -- this mode keeps the reference's exact I.
#ifdef SUCRE_EXP_LIGHT_VECTOR_BASES   // (A/B: the strip's bases as the compiler holds them, not forced into scalar registers)
constexpr bool kExpLightVectorBases = true;
#else
constexpr bool kExpLightVectorBases = false;
#endif
// light.hip: workgroups of a gradient launch (0: as many as are resident).  Round 6, light + closed form, whose second pass re-reads
// what the first read (a wave's strip is 83 KB at 65 levels; 4096 waves: 340 MB against the 256 MB memory-side cache): see HISTORY 14.
#ifndef SUCRE_EXP_LIGHT_GRID
#define SUCRE_EXP_LIGHT_GRID 0
#endif
#ifdef SUCRE_EXP_LIGHT_LOAD_Z   // (A/B: the J-parameter light kernel loads the stored ranges, as until round 6, instead of forming ||cP||)
constexpr bool kExpLightLoadZ = true;
#else
constexpr bool kExpLightLoadZ = false;
#endif
#ifndef SUCRE_LIGHT_FOLD1
#define SUCRE_LIGHT_FOLD1 0
#endif
#ifndef SUCRE_LIGHT_FOLD2
#define SUCRE_LIGHT_FOLD2 0
#endif
constexpr bool kExpLightFold1 = SUCRE_LIGHT_FOLD1 != 0, kExpLightFold2 = SUCRE_LIGHT_FOLD2 != 0;

// fit.hip, timing only: every wave of a J-parameter launch records when it entered and left its strips (100 MHz wall clock) --
// how much of a launch is its ragged end (tools/exp/wave_times.py).
#ifdef SUCRE_EXP_WAVE_TIMES
constexpr bool kExpWaveTimes = true;
#define SUCRE_EXP_EXPORT extern "C" __attribute__((visibility("default")))
#else
constexpr bool kExpWaveTimes = false;
#define SUCRE_EXP_EXPORT [[maybe_unused]] static
#endif

// fit.hip: s_setprio experiments against the hardware's oldest-wave-first issue order (tools/exp/wave_times.py shows a launch's
// five resident workgroups per CU finishing one after the other).  0: none; 1: a wave's priority rotates with every strip,
// offset by the workgroup's generation; 2: static, younger workgroups higher; 3: rotates with every item.
#ifndef SUCRE_EXP_PRIO
#define SUCRE_EXP_PRIO 0
#endif
constexpr int kExpPrio = SUCRE_EXP_PRIO;

// layout.h: the work a wave of each workgroup generation is dealt, in 64ths of what a wave of generation 0 (the oldest) gets;
// the first must be 64.  64 everywhere = equal shares.  Measured (tools/exp/wave_times.py, ab_solo.sh, round 4): with
// 64,44,24,14,5 a J-parameter launch ALONE on the GPU takes 126 instead of 134 us (its waves then end within 20 us of each
// other instead of 75), but with two images in flight -- the default -- the second launch's workgroups arrive as the first one's
// leave, no longer one generation per CU, and an image takes 25.1 instead of 24.4 ms.  Round 5 (tools/exp/ab_vs.sh, one box,
// two rounds each; launch alone / Mpix/s with two images in flight): equal shares 131.9-132.1 us / 84.91-84.95;
// **64,50,38,28,20: 127.3-128.0 / 84.89-85.06** (the product: nothing lost in flight, 3.3 % gained alone);
// 64,56,48,40,32: 131.3 / 84.3; 64,48,36,28,22: 128.1-128.7 / 84.5-84.7; 64,52,40,28,16: 125.7-125.9 / 84.54-84.57;
// 64,48,34,22,12: 125.2-125.6 / 84.2-84.6; 64,46,30,20,10: 125.2-125.5 / 84.3; 64,44,24,14,5: 126.2-126.6 / 81.4-81.5.
#ifndef SUCRE_DEAL_FIT
#define SUCRE_DEAL_FIT 64, 50, 38, 28, 20
#endif
// The closed-form kernel (four generations, instruction-bound) takes the unequal deal as its product default (round 5): alone a
// launch takes 140.8-142.3 instead of 148.4 us (same box, tools/exp/ab_vs.sh; 0.51 instead of 0.49 of the HBM peak by SURVEY
// 8(d)'s bytes) and an image restored alone 29.4 instead of 30.9 ms; with two images in flight 75.6-75.9 against 75.9 Mpix/s --
// unlike the J-parameter kernel it loses nothing there.  (64,54,44,34: 143.5 us; 64,44,28,12: 139-140 us alone but 75.0 Mpix/s
// in flight; 64,56,40,24: 140.9.)
#ifndef SUCRE_DEAL_CLOSED
#define SUCRE_DEAL_CLOSED 64, 48, 32, 20
#endif

// fit.hip batch launches, timing only (results meaningless): where an image's time inside a batch launch of 640x480 images
// goes.  1: the tail launch reduces the groups but nobody takes the totals or steps; 2: no pass (every wave is told it has no
// strip: descriptors, wave sums and the hand-in alone).  tools/exp/batch_ablation.sh.
#ifndef SUCRE_EXP_BATCH
#define SUCRE_EXP_BATCH 0
#endif
constexpr int kExpBatch = SUCRE_EXP_BATCH;

// fit.hip, closed-form kernel: a chunk's 24 exponentials in two batches of twelve (two levels each) instead of one of 24 --
// twelve registers fewer in flight, which is what the kernel lacks to fit five waves per SIMD (SUCRE_CLOSED_WAVES=5).
// Measured (round 5, same box, tools/exp/ab_vs.sh, config 2, launch alone): product 146.5-147.9 us; two batches at 4 waves per
// SIMD 152-153; at 5 waves with the unequal deal 150.5; at 5 waves with equal shares 158-160.  Neither the smaller batches nor
// the fifth wave pays: the 24 exponentials back to back at four waves stay.
#ifdef SUCRE_EXP_HALF_EXPS
constexpr bool kExpHalfExps = true;
#else
constexpr bool kExpHalfExps = false;
#endif
// ... and for such occupancy experiments the closed-form instantiation of batch_iter_kernel (which needs more registers than
// five waves leave) compiled to nothing.
#ifdef SUCRE_EXP_NO_BATCH_CLOSED
constexpr bool kExpNoBatchClosed = true;
#else
constexpr bool kExpNoBatchClosed = false;
#endif

// match.hip, timing only (the store is then garbage): match_kernel without its dense-chunk stores -- what the first pass of a
// count-first / write-sorted matching (VERDICT r04 task 7: trade scatter_kernel's read-back for a second matching pass) would
// cost at least: every projection, gather and ballot, no observation written.  tools/exp/ab_lib.sh countonly match_kernel.
#ifdef SUCRE_EXP_MATCH_COUNT_ONLY
constexpr bool kExpMatchCountOnly = true;
#else
constexpr bool kExpMatchCountOnly = false;
#endif

// fit.hip, batch launches: a wave's streams of consecutive images chained into one (StreamChain).  0 = every image's stream on
// its own (trailing copies, drain, prime), the form of the first batch kernel -- for the same-box A/B.
#ifndef SUCRE_EXP_BATCH_CHAIN
#define SUCRE_EXP_BATCH_CHAIN 1
#endif
constexpr bool kExpBatchChain = SUCRE_EXP_BATCH_CHAIN != 0;

// J-parameter kernels, timing only: the stepped state of every strip written to ONE place per wave (wrong results): what of
// the stores' cost is their issue and acknowledgement, and what the write stream to HBM.
#ifdef SUCRE_EXP_STORE_LOCAL
constexpr bool kExpStoreLocal = true;
#else
constexpr bool kExpStoreLocal = false;
#endif

// wave_sums through __shfl_down (ds_bpermute) as until round 5, for the same-box A/B of the register-to-register form.
#ifdef SUCRE_EXP_SHFL_SUMS
constexpr bool kExpShflSums = true;
#else
constexpr bool kExpShflSums = false;
#endif
// ... or one register-to-register tree per sum (round 5's form) instead of round 6's several sums to a register: same bits, 80 against 38 instructions.
#ifdef SUCRE_EXP_PLAIN_WAVE_SUMS
constexpr bool kExpPlainWaveSums = true;
#else
constexpr bool kExpPlainWaveSums = false;
#endif
