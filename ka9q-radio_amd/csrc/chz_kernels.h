// chz_kernels.h -- hand-written CDNA4 (gfx950) kernels of the overlap-save channelizer.
//
// What the reference does on CPU threads with FFTW3 (src/filter.c):
//   K1  forward transform of the N-sample window          src/filter.c:505-508,573-582
//   K2  spur notches on a handful of bins                  src/filter.c:464-474 (ordered across blocks by a device ticket: folded into
//                                                            fwd_rows for the usual short lists, the notch_fix kernel otherwise)
//   K3  per-channel bin gather x frequency response        src/filter.c:728-911
//   K4  per-channel small backward transform, keep olen    src/filter.c:914, :357
// is done here by four kernels.  The large transform is a three-axis Cooley-Tukey
// decomposition N = Na*Nb*Nc (n = na*Nb*Nc + nb*Nc + nc, k = ka + Na*kb + Na*Nb*kc):
//
//   fwd_first_real   axis a.  Real input: two adjacent real columns travel as one
//                    complex column, the Hermitian split happens inside the tile, so
//                    only rows ka = 0..Na/2 are ever written.  Reads the input ring.
//   fwd_cols         axis b (and axis a of a complex-input master): column FFTs in
//                    place, T adjacent columns per workgroup.
//   fwd_rows         axis c.  Row FFTs; the tile is transposed through LDS so the
//                    digit-reversed store k = ka + Na*(kb + Nb*kc) is contiguous in ka,
//                    and the upper half of each row lands conjugated at N-k.
//
// Each workgroup keeps its tile in VGPRs, does one radix-R1 and one radix-R2
// butterfly layer (regfft.h) with a single LDS exchange between them ("LDS-staged
// radix butterflies"); inter-axis twiddles W^(k*m) are factored as
// W^(k*tile_base) * W^(k*offset_in_tile), two tiny L2-resident tables built in
// float64 on the host.  No MFMA: this is an HBM/L2-bound permutation-heavy
// transform, not a dense contraction.
//
// The file is plain HIP, gfx950 only; tests/hipemu can also run it on the CPU as
// test infrastructure (it is NOT a fallback: the product library is hipcc-built).
#pragma once
#include <hip/hip_runtime.h>
#include "regfft.h"

// Rounding fences.  The library is built with -ffp-contract=fast, which lets the backend fuse ANY multiply
// with a following add, pragmas notwithstanding; where a result must round like the reference's x86-64
// build (no FMA) the product goes through an empty asm so the two operations cannot be combined.
#if defined(__HIP_DEVICE_COMPILE__)
#define CHZ_ROUNDED_F32(x) asm volatile("" : "+v"(x))
#define CHZ_ROUNDED_F64(x) asm volatile("" : "+v"(x))
#else
#define CHZ_ROUNDED_F32(x) ((void)0)
#define CHZ_ROUNDED_F64(x) ((void)0)
#endif

// Wavefront-level ordering point for LDS traffic between lanes of ONE wavefront: free on the GPU (lanes run in
// lockstep and LDS operations of a wavefront complete in order); the CPU emulator runs lanes one after another
// and needs a real rendezvous.
#if defined(__HIP_DEVICE_COMPILE__)
#define CHZ_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#elif defined(HIPEMU)
#define CHZ_WAVE_SYNC() hipemu_wave_barrier(hipemu_linear_tid())
#else
#define CHZ_WAVE_SYNC() ((void)0)      /* hipcc's host pass only parses the kernels */
#endif

namespace chz {

// Stores of the forward passes.  A pass's output is read next by another kernel on whatever XCD it lands on, so a
// plain store only parks a dirty line in this XCD's (non-coherent) L2 until the end-of-kernel release writes all of
// them back in one burst -- measured as ~2 us of every pass.  Stores with agent scope (the `sc1` bit) write through as
// the kernel runs and keep the lines cacheable for the next pass:
//     config 3, one MI355X:  fwd_first_real 10.5 -> 8.2 us, fwd_cols 8.4 -> 7.0 us, fwd_rows 7.6 -> 6.7 us,
//     pipelined block time 16.9 -> 15.4 us.
// Variants measured and rejected: non-temporal stores (`nt`: the next pass then misses the cache, pipelined 17.9 -> 19.3 us)
// and relaxed agent-scope atomic stores from C++ (8-byte pieces, more instructions: pipelined worse).  CHZ_WT=0 at build
// time restores plain stores.
#ifndef CHZ_WT
#define CHZ_WT 1
#endif
// The stores go through a buffer descriptor (raw_buffer_store with aux = 16 = sc1) rather than inline asm, so the compiler
// still counts and schedules them and pads their hazards; offsets are 32-bit bytes from the (wave-uniform) array base.
#if defined(__HIP_DEVICE_COMPILE__) && CHZ_WT
typedef unsigned chz_u2 __attribute__((ext_vector_type(2)));
typedef unsigned chz_u4 __attribute__((ext_vector_type(4)));
#define CHZ_OUT_DESC(name, base) const __amdgpu_buffer_rsrc_t name = __builtin_amdgcn_make_buffer_rsrc((void*)(base), 0, 0x7ffffffc, 0x00020000)
__device__ __forceinline__ void store_wt(__amdgpu_buffer_rsrc_t d, int elem, float2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(chz_u2{__float_as_uint(v.x), __float_as_uint(v.y)}, d, elem * 8, 0, 16);
}
__device__ __forceinline__ void store_wt(__amdgpu_buffer_rsrc_t d, int elem, float4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(chz_u4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, d, elem * 16, 0, 16);
}
// (round 6) chan_ifft's STAGED output rows -- launches of >= 16384 channels: at C_rt 33 GB per block that nobody on the device reads again soon -- leave as
// NON-TEMPORAL stores (CHZ_CHAN_NT bit 1, shipped).  Measured at 20.0 M channels (profiles/r06_chan_nt.txt): mean block 19.20-19.41 -> 18.77-18.78 ms, the
// C_rt search 19.75 M -> 20.50 / 20.75 M sustained on the same box, the 8f chain at 1.5 M channels 4.20-4.26 -> 4.15-4.16 ms; PCM / outputs bit-identical.
// Bit 0 -- the response rows (41 GB per block, read once) as non-temporal LOADS -- changes nothing (19.23-19.34 ms) and stays off; both together 18.82-18.88.
// Small launches keep plain stores (their outputs are read back or demodulated out of the caches).  A/B builds: make ../libchz_hip_cnt0.so | _cnt1.so | _cnt3.so.
#ifndef CHZ_CHAN_NT
#define CHZ_CHAN_NT 2
#endif
typedef float chz_f2v __attribute__((ext_vector_type(2)));
typedef float chz_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float2 load_resp(const float2* p) {
  if constexpr ((CHZ_CHAN_NT & 1) != 0) { const chz_f2v r = __builtin_nontemporal_load(reinterpret_cast<const chz_f2v*>(p)); return make_float2(r.x, r.y); }
  else return *p;
}
__device__ __forceinline__ void store_out4(float4* p, float4 v) {
  if constexpr ((CHZ_CHAN_NT & 2) != 0) { const chz_f4v r = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(r, reinterpret_cast<chz_f4v*>(p)); }
  else *p = v;
}
// gathers through a buffer descriptor: a 32-bit element offset from a wave-uniform base instead of a 64-bit address per lane
#define CHZ_IN_DESC(name, base) const __amdgpu_buffer_rsrc_t name = __builtin_amdgcn_make_buffer_rsrc((void*)(base), 0, 0x7ffffffc, 0x00020000)
__device__ __forceinline__ float2 load_f2(__amdgpu_buffer_rsrc_t d, int elem) {
  const chz_u2 r = __builtin_amdgcn_raw_buffer_load_b64(d, elem * 8, 0, 0);
  return make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
}
#define CHZ_LOAD2(desc, base, elem) load_f2(desc, (int)(elem))
__device__ __forceinline__ float load_f1(__amdgpu_buffer_rsrc_t d, int elem) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d, elem * 4, 0, 0));
}
#define CHZ_LOAD1(desc, base, elem) load_f1(desc, (int)(elem))
#define CHZ_STORE(desc, base, elem, value) store_wt(desc, (int)(elem), (value))
// predicated store without a branch: a raw buffer access past num_records is dropped by the hardware
#define CHZ_STORE_IF(desc, base, elem, value, cond) store_wt(desc, (cond) ? (int)(elem) : 0x10000000, (value))
#define CHZ_LOAD_RESP(ptr) load_resp(ptr)
#define CHZ_STORE_OUT4(ptr, value) store_out4((ptr), (value))
#else
#define CHZ_OUT_DESC(name, base) const int name = 0; (void)name
#define CHZ_IN_DESC(name, base) const int name = 0; (void)name
#define CHZ_LOAD_RESP(ptr) (*(ptr))
#define CHZ_STORE_OUT4(ptr, value) (*(ptr) = (value))
#define CHZ_LOAD2(desc, base, elem) ((base)[(elem)])
#define CHZ_LOAD1(desc, base, elem) ((base)[(elem)])
#define CHZ_STORE(desc, base, elem, value) ((base)[(elem)] = (value))
#define CHZ_STORE_IF(desc, base, elem, value, cond) do { if (cond) (base)[(elem)] = (value); } while (0)
#endif

// ------------------------------------------------------------------------------
// parameter blocks (plain data; filled by chz_engine.hip)
// ------------------------------------------------------------------------------
struct FirstRealParams {
  const float* ring;      // input sample ring (device), float32
  long ring_len;          // ring length in floats (even)
  long start;             // window start inside the ring (even)
  float2* buf;            // out: [Ra][inner] complex
  int inner;              // Nb*Nc  (even)
  int T;                  // packed (complex) columns per tile; inner/2 % T == 0
  int Ra;                 // Na/2 + 1 rows kept
  int padk;               // LDS padding (float2) after each k1 group
  const float2* tw_sub;   // [R2][R1]   W_Na^(j*k1)
  const float2* tw_tile;  // [tiles][Ra] W_N^(ka * 2*c0)
  const float2* tw_col;   // [Ra][2T]   0.5 * W_N^(ka*cc) (even cc) or -0.5i * W_N^(ka*cc) (odd cc)
  // raw A/D samples.  When ring16 != nullptr the ring holds int16 (same index space as
  // `ring`) and each sample becomes (float)x * scale16 on load -- the reference's convert(),
  // src/rx888.c:753-767, incl. the LTC2208 de-randomiser (if bit 0 is set flip bits 1..15, :711-716).
  const short* ring16;
  float scale16;
  int derand;
  // per-wavefront partial sums over the block's NEW samples (window index >= new_from = M-1):
  // sum of x*x (frontend->if_power, :780-795) and the count of |x| > 32766 (frontend->overranges)
  unsigned long long* energy_part;   // [grid * waves] or nullptr
  unsigned* clip_part;               // [grid * waves]
  int new_from;
#if CHZ_FWD_BATCH
  int nbatch; long bstart[4]; float2* bbuf[4];     // blockIdx.y picks the block's window start and intermediate buffer
#endif
};

// A/B of the north star's "wavefront-shuffle twiddles" (build with -DCHZ_TW_SHUFFLE=1; tiles of T = 16 columns only): the epilogue's
// column factors W_N^(ka*cc) are not read from tw_col per lane but generated across each row of 16 lanes -- one broadcast load of
// the row's base factor, then powers handed from lane to lane in four DPP steps.  Measured slower (profiles/r03_twiddle_ab.jsonl: 8.0 -> 9.3 us
// for the pass, pipelined block 14.4 -> 15.3 us); a compile-time switch, because even an untaken run-time branch around it cost the pass 1.1 us.
#ifndef CHZ_TW_SHUFFLE
#define CHZ_TW_SHUFFLE 0
#endif
// value of lane (lane - n) inside the lane's row of 16 (DPP row_shr:n on the device, a plain shuffle on the CPU test emulator)
template <int N_> __device__ __forceinline__ float row_shr16(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x110 + N_, 0xf, 0xf, false));
#else
  return __shfl_up(v, (unsigned)N_);
#endif
}

// EXPERIMENT (round 4, build variant `make xcdaffine`, -DCHZ_XCD_AFFINE=1; the shipped library carries none of it).  The one
// byte-saving idea left for the forward transform: make fwd_rows read what fwd_cols wrote out of the SAME XCD's L2.  fwd_rows'
// workgroup (kb, a-tile) reads rows ka of its a-tile at one kb over all nc; fwd_cols' workgroup (ka, column tile) writes every kb of
// its columns -- so producers and consumers form one connected component per a-tile (5 for config 3: ka in [16c - 12, 16c + 4)).
// Placement: the dispatcher is observed to put block b on XCD b % 8 (MI355X_MICROARCH.md), so both passes launch 8 x (work per
// component) blocks, block b serves component (b % 8 - rot) mod 8 and leaves at once if that component does not exist; `rot`
// advances by ncomp per block so consecutive blocks in flight load different XCDs.  fwd_cols then stores PLAIN (the lines stay
// in its XCD's L2; write-through stores drop them).  Decision record: DESIGN.md section 7, profiles/r04_xcd_affine.txt.
// Wavefronts per SIMD the lane-per-channel passes are compiled for (the second __launch_bounds__ argument is HIP's minimum number of
// wavefronts per execution unit).  Both want ~235 VGPRs; at 3 wavefronts the compiler gets 168 and spills the rest to scratch.  Measured
// at 1.5 M channels (round 4): demod_fm_lanes 2.28 -> 2.07 ns per channel with 3 (its walks are latency-bound, the spills sit outside the
// per-sample chains), pll_lanes 2.52 -> 3.02 ns (its spills land inside the loop): FM ships 3, the PLL pass stays at 2.
#ifndef CHZ_PLL_WAVES
#define CHZ_PLL_WAVES 2
#endif
#ifndef CHZ_LIN_WAVES
#define CHZ_LIN_WAVES 2             // demod_lin_lanes: the compiler takes 165 VGPRs with this bound = 3 wavefronts per SIMD (A/B build: -DCHZ_LIN_WAVES=4 caps it at 128)
#endif
#ifndef CHZ_PLL_UNROLL
#define CHZ_PLL_UNROLL 1            // pll_lanes: samples read / stepped / written per group (A/B build: make pllu4)
#endif
#ifndef CHZ_LIN_UNROLL
#define CHZ_LIN_UNROLL 4            // demod_lin_lanes' final pass: samples read / stepped / written per group (1 = round 4's loop; measured at 1.5 M channels on one
                                    // box, every stream on its own queue: 1 -> 4.28, 4 -> 4.07, 8 -> 4.06 ms per block, PCM bit-identical; A/B: make linu1 / linu8)
#endif
#ifndef CHZ_LIN_PACKED_STORE
#define CHZ_LIN_PACKED_STORE 1      // demod_lin_lanes: mono S16 rows leave as 8-byte words (A/B build: -DCHZ_LIN_PACKED_STORE=0)
#endif
#ifndef CHZ_FM_WAVES
#define CHZ_FM_WAVES 3
#endif
#ifndef CHZ_FM_DISC_UNROLL
#define CHZ_FM_DISC_UNROLL 1        // demod_fm_lanes: samples whose discriminator phases (double atan2) are computed side by side.  Measured at 1.5 M channels
                                    // (round 6, profiles/r06_fm_disc_unroll.txt; A/B builds: make ../libchz_hip_fmd2.so, _fmd4.so, _fmd8.so): 1 -> 1.93-1.98 ns per
                                    // channel, 2 -> 1.99-2.01, 4 -> 2.13-2.17, 8 -> 2.49-2.50 (4 at 2 wavefronts per SIMD, no spills: 1.91-1.93): the pass is
                                    // bound by vector ISSUE (~300 instructions per sample, a third of them the atan2), not by the latency of its chains --
                                    // overlapping the chains buys nothing and the extra live registers cost spills.  Ships 1 = round 5's loop.
#endif
#ifndef CHZ_XCD_AFFINE
#define CHZ_XCD_AFFINE 0
#endif
// EXPERIMENT build (round 5, -DCHZ_FWD_BATCH=1, ../libchz_hip_batch.so): the three forward passes of 2 or 4 CONSECUTIVE blocks as one grid
// each (blockIdx.y = the block within the batch): 4 x the wavefronts per launch, a third of the launches.  Decision record: DESIGN.md section 7.
#ifndef CHZ_FWD_BATCH
#define CHZ_FWD_BATCH 0
#endif
// (the block's pointers are picked with compile-time indices: writing to the by-value parameter struct, or indexing its arrays with
//  blockIdx.y, sends the whole struct through scratch memory -- measured: every pass 3.7 x slower, batched or not)
#if CHZ_FWD_BATCH
#define CHZ_BSEL(arr, dflt) (p.nbatch > 1 ? (blockIdx.y == 0 ? arr[0] : blockIdx.y == 1 ? arr[1] : blockIdx.y == 2 ? arr[2] : arr[3]) : (dflt))
#else
#define CHZ_BSEL(arr, dflt) (dflt)
#endif
struct XcdAffine { int on, Ta, shift, rot, ncomp; };

struct ColsParams {
  const float2* in;       // in[(row*NP + n)*inner + col]  (+ ring wrap when in_len != 0)
  long in_len;            // 0, or ring length in float2 (input then starts at in_start)
  long in_start;
  float2* out;            // out[(row*NP + k)*inner + col]; may alias in when in_len == 0
  int rows;               // independent row blocks (ka values)
  int inner;              // contiguous inner length (columns per row block)
  int T;                  // columns per tile; inner % T == 0
  int padk;
  const float2* tw_sub;   // [R2][R1]      W_NP^(j*k1)
  const float2* tw_tile;  // [inner/T][NP] W_(NP*inner)^(k * c0)
  const float2* tw_col;   // [NP][T]       W_(NP*inner)^(k * t)
  const float2* tw_full;  // [NP][inner] W_(NP*inner)^(k * col), or nullptr: one load and no product per output (axis b: the table
                          // is Nb*Nc entries and L2-resident; axis a of a complex master would need N entries and keeps the two factors)
  XcdAffine xa;           // experiment build only (CHZ_XCD_AFFINE): XCD-affine placement of the axis-b pass
#if CHZ_FWD_BATCH
  int nbatch; float2* bbuf[4];
#endif
};

// Spectrum storage: bin k = ka + Na*x lives at  spec[x*pitch + off + ka].  pitch = Na, off = 0 is
// the natural order.  For a real master with odd Na the planner picks pitch = a multiple of 16 and
// an offset such that BOTH the direct 16-bin store segments and the conjugate-mirrored ones start on
// 128-byte lines (misaligned 128-byte segments cost ~2x on this chip).
struct SpecLayout { int na, pitch, off; };
__host__ __device__ __forceinline__ long spec_addr(const SpecLayout& l, long k) {
  const long x = k / l.na;
  return x * l.pitch + l.off + (k - x * l.na);
}

// storage index of master bin k without a division: k + (k / na) * (pitch - na) + off
__device__ __forceinline__ int spec_index(int na_off, unsigned magic, int dpitch, int k) {
  const unsigned q = __umulhi((unsigned)k, magic);
  return k + (int)__umul24(q, (unsigned)dpitch) + na_off;
}

// K2 inside fwd_rows (apply_notch_filters, src/filter.c:464-474).  Every listed bin is stored by exactly ONE thread of ONE workgroup
// of this pass, and that thread has the bin's value in a register: the host names it (workgroup, thread, output index K2 -- notch_owner()
// in chz_launch.h restates the kernel's own index arithmetic), the thread takes the block's ticket, runs the recurrence on its register
// and stores the notched value.  No kernel of its own (3.6 us of queue time per block on the 4-stream trace of round 3), no second
// trip of the bin through memory.  Ticket / tombstone / error-word semantics are those of notch_fix (below); only lists that fit the
// kernel arguments (CHZ_NOTCH_INLINE entries: radiod's DC-only or few-spur lists) ride here, ordered by the device ticket.
#define CHZ_NOTCH_INLINE 8
struct NotchOwn { int wg, tid, k2, pad; };   // who stores the entry's bin: workgroup (blockIdx.x), thread, second-layer output index
struct RowsNotch {
  int n;                            // list entries (0: nothing folded into this launch)
  int nwg;                          // distinct owner workgroups: the last one to finish publishes the ticket
  int wg[CHZ_NOTCH_INLINE];         // owner workgroup per entry, in the kernel arguments (read with compile-time indices only: every
                                    // workgroup of the pass asks "is it me?" with eight scalar compares and no memory access)
  // the rest of the list lives in device memory, touched by the owner workgroups only
  const NotchOwn* own;              // [n]
  const int* addr;                  // [n] storage index of the entry's bin: checked against what the thread is about to store
  const int* next;                  // [n] next entry naming the same bin, or -1
  const int* head;                  // [n] 1 if the entry is the first one naming its bin
  const double* alpha;              // [n]
  double* state;                    // [n][2] persists across blocks
  unsigned* ver;                    // [4] ticket counter, tombstone, graph base, owner-workgroup count of the block in progress
  unsigned seq;                     // this block's ticket (relative to *seq_base when that is set: captured launches)
  unsigned* seq_base;
  unsigned adv;                     // != 0: last block of a captured sequence: moves *seq_base on
  unsigned* err;                    // host-visible error word
  long long max_wait;               // ticket wait budget, ticks of the constant-rate counter
};

struct RowsParams {
  const float2* buf;      // [Ra][Nb][Nc]
  float2* spec;           // out: master spectrum in SpecLayout order
  SpecLayout lay;
  int Ra, Na, Nb;         // rows kept, axis-a length, axis-b length
  int Ta;                 // ka values per tile
  int ka_shift;           // tile t covers ka in [t*Ta - ka_shift, (t+1)*Ta - ka_shift)
  int ld, padg;           // LDS leading dimension (Ta+1) and per-group padding
  long N;                 // full transform length
  int mirror;             // 1: real master (bins N/2+1, conj-mirror store); 0: complex master
  const float2* tw_sub;   // [R2][R1] W_Nc^(j*k1)
  XcdAffine xa;           // experiment build only (CHZ_XCD_AFFINE)
  RowsNotch nf;           // K2 folded into this pass (nf.n == 0: none; the notch_fix kernel follows instead, or there is no list)
#if CHZ_FWD_BATCH
  int nbatch; float2* bbuf[4]; float2* bspec[4];
#endif
};

// One channel's gather, precomputed on the host from `shift`
// (restating the index walk of src/filter.c:728-911):
// the t-th output bin counted from the most negative frequency takes master bin
//   src0 + dir*(t - t0)   (mod wrap if wrap != 0)     for t0 <= t < t0 + cnt
// and is zero otherwise; conj != 0 conjugates the master bin (inverted spectrum).
// row   = which row of the bank's response array holds this channel's response (set_filter swaps a response by
//         writing a spare row and re-pointing the descriptor, so blocks in flight keep the old one);
// shift = the channel's bin shift itself (REAL-output gather, noise estimate).
// Descriptors exist once per spectrum slot: a retune edits the copy of the NEXT block of each slot in stream
// order and never touches what blocks already in flight are reading.
struct ChanDesc { int t0, cnt, src0, dir, conj, wrap, row, shift; };

// Fine tuning of one channel (the tail of downconvert(), src/radio.c:1476-1520), stateless in the block
// number so blocks in flight on different streams cannot race on an oscillator state.  With
// kb = job - job0 blocks and g = kb*olen + m samples since the base, output sample m of block `job` is
// rotated by  phase0 + ((kb+1)*adj_num mod V)/V + g*freq + rate*g*(g+1)/2   cycles:
//   freq/rate  = what set_osc() was given (cycles/sample, cycles/sample^2; src/osc.c:28-47, :60-70),
//   adj_num/V  = the per-block phase_adjust = cispi(2*(shift % V)/V)   (src/radio.c:1493,1497),
//   phase0     = everything accumulated before the base, including the shift-change kick (src/radio.c:1494).
struct FineDesc { double phase0, freq, rate; unsigned job0; int adj_num, V, on;
                  // for the register-tiled channel kernel, computed by the host when the channel is (re)tuned: job0 mod V, and the rotator's
                  // advance over `stride` samples (the kernel's lanes hold every stride-th sample: stride = its R1), e^{2 pi i stride freq}
                  int job0m, stride; double step_c, step_s; };

// slave->beam (src/filter.c:756-775): Y = (alpha X[rp] + beta conj(X[m_bins-rp])) H, weights as set_filter_weights stores them
struct BeamDesc { double ar, ai, br, bi; int on; int pad; };

struct ChanParams {
  int m_bins, m_real;     // master bins; master is REAL (else COMPLEX)
  int stage;              // 1: output rows leave through LDS as full-line stores (throughput); 0: straight from the lanes (latency)
  const unsigned char* isb; // [nch] or nullptr: per-channel slave->isb flags (EPI variant only)
  const BeamDesc* beam;   // [nch] or nullptr: slave->beam with its weights (EPI variant only, COMPLEX masters)
  const FineDesc* fine;   // [nch] or nullptr: plain execute_filter_output semantics
  double* power;          // [nch] mean |sample|^2 of the block after rotation (chan->sig.bb_power, :1516-1520)
  // demod_linear()'s AGC looks at the block once before it demodulates it: the largest energy of a 2 ms slice (src/linear.c:177-203).
  // With the output rows staged through LDS (p.stage) that first look happens HERE, on the samples still in LDS, in the reference's
  // own order -- and the lane-per-channel demodulator reads the baseband once instead of twice (1920 of 8.9 KB per 12 kHz channel).
  double* agc_peak;       // [nch] or nullptr (EPI variant, stage mode only)
  int agc_sps;            // samples per slice: rint(olen * .002 / blocktime), at least 1
  unsigned job;
  // the block phase correction's integers without a division in the kernel (round 4): V = the master's overlap factor (the same for
  // every channel), 1/V, job mod V and 2^32 mod V, by the host per launch
  int fine_V; double fine_invV; int fine_jobm, fine_wrapm;
  const float2* spec;     // master spectrum of this block (SpecLayout order)
  SpecLayout lay;
  float inv_na;           // 1/na, for the bin -> (row, column) split
  unsigned magic;         // ceil(2^32 / na): bin / na by one multiply-high (exact for bins < 2^32 / na, checked by chan_layout)
  int dpitch;             // pitch - na
  const float2* resp;     // [rows][P] frequency responses, row = desc[ch].row
  const ChanDesc* desc;   // [nch] this slot's descriptors
  float2* out;            // [nch][olen]
  int ch0, nch, olen;     // channels [ch0, ch0+nch) of the bank are processed
  const float2* tw_sub;   // [R2][R1]  W_P^(-j*k1)  (backward)
};

// ------------------------------------------------------------------------------
// K1a: first axis of a REAL master.  grid = inner/2/T tiles.
// ------------------------------------------------------------------------------
template <int R1, int R2>
__global__ void fwd_first_real(FirstRealParams p) {
  constexpr int NA = R1 * R2;
  constexpr int LA = R1 > R2 ? R1 : R2;
  // rows one lane walks in the split epilogue: ceil(Ra / rows-per-sweep), rows-per-sweep >= LA
  constexpr int NE = (NA / 2 + 1 + LA - 1) / LA;
  HIP_DYNAMIC_SHARED(float2, lds)
  const int tid = threadIdx.x, nthr = blockDim.x, tile = blockIdx.x;
  const int T = p.T;
  const int c0 = tile * T;                       // first packed column of the tile
  const float2* __restrict__ ring2 = reinterpret_cast<const float2*>(p.ring);
  const long ring2_len = p.ring_len >> 1, start2 = CHZ_BSEL(p.bstart, p.start) >> 1, inner2 = p.inner >> 1;
  const float2* __restrict__ tws = p.tw_sub;
  const float2* __restrict__ twt = p.tw_tile;
  const float2* __restrict__ twc = p.tw_col;
  float2* __restrict__ gout = CHZ_BSEL(p.bbuf, p.buf);
  float2* zl = lds + NA * T + R1 * p.padk;         // second LDS region (see layer 2)

  // Epilogue geometry is known up front: one lane owns one PACKED column pc (= two adjacent real
  // columns 2pc, 2pc+1) and walks rows kr, kr+rpi, ...  Its twiddle factors are fetched now as one
  // 16-byte load per row; they are consumed at the very end, where each row becomes one 16-byte store.
  const int rpi = nthr / T;                      // rows covered per sweep (>= LA)
  const int kr = tid / T, pc = tid - kr * T;
  const bool lane_on = kr < rpi;
  float2 we1[NE];
  float4 we2[NE];
  static_for<NE>([&](auto u) {
    constexpr int U = decltype(u)::value;
    const int k = kr + U * rpi;
    const int kk = (lane_on && k < p.Ra) ? k : 0;
    we1[U] = twt[tile * p.Ra + kk];
#if !CHZ_TW_SHUFFLE
    we2[U] = reinterpret_cast<const float4*>(twc)[kk * T + pc];   // columns 2pc (even) and 2pc+1 (odd)
#else
    {
      // the row's two base factors (columns 0 and 1: 1/2 and h = -i/2 W_N^ka), the same address in all 16 lanes of the row
      const float4 b = reinterpret_cast<const float4*>(twc)[kk * T];
      const float2 h = make_float2(b.z, b.w);
      const float2 w = make_float2(-2.f * h.y, 2.f * h.x);          // W_N^ka = 2i h
      float2 g = cmul(w, w);                                        // ratio between neighbouring lanes: W_N^(2 ka)
      float2 x = make_float2(1.f, 0.f);                             // becomes g^pc
      { const float2 q = make_float2(row_shr16<1>(x.x), row_shr16<1>(x.y)); if (pc & 1) x = cmul(q, g); }
      g = cmul(g, g);
      { const float2 q = make_float2(row_shr16<2>(x.x), row_shr16<2>(x.y)); if (pc & 2) x = cmul(q, g); }
      g = cmul(g, g);
      { const float2 q = make_float2(row_shr16<4>(x.x), row_shr16<4>(x.y)); if (pc & 4) x = cmul(q, g); }
      g = cmul(g, g);
      { const float2 q = make_float2(row_shr16<8>(x.x), row_shr16<8>(x.y)); if (pc & 8) x = cmul(q, g); }
      const float2 od = cmul(h, x);
      we2[U] = make_float4(0.5f * x.x, 0.5f * x.y, od.x, od.y);
    }
#endif
  });

  unsigned long long energy = 0; unsigned clips = 0;        // int16 input only
  // layer 1: radix R1 over na = j + q*R2, one (j, column) pair per thread
  if (tid < R2 * T) {
    const int j = tid / T, t = tid - j * T;
    float2 v[R1], w1[R1];
    long f64 = start2 + (long)j * inner2 + c0 + t;
    if (f64 >= ring2_len) f64 -= ring2_len;
    const int first = (int)f64, len = (int)ring2_len, qstep = R2 * (int)inner2;   // 32-bit from here on
    // only the windows that straddle the end of the ring need the wrap test (uniform per workgroup)
    const bool nowrap = start2 + c0 + T - 1 + (long)(NA - 1) * inner2 < ring2_len;
    if (p.ring16 == nullptr && nowrap) {
      const float2* __restrict__ g0 = ring2 + first;
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        v[Q] = g0[Q * qstep];
        if constexpr (Q > 0) w1[Q] = tws[j * R1 + Q];
      });
    } else if (p.ring16 == nullptr) {
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        int idx = first + Q * qstep;                 // < 2*len: the window is shorter than the ring
        if (idx >= len) idx -= len;
        v[Q] = ring2[idx];
        if constexpr (Q > 0) w1[Q] = tws[j * R1 + Q];
      });
    } else {
      const unsigned* __restrict__ ring16 = reinterpret_cast<const unsigned*>(p.ring16);   // two samples per word
      unsigned raw[R1];
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        int idx = first + Q * qstep;
        if (idx >= len) idx -= len;
        raw[Q] = ring16[idx];
        if constexpr (Q > 0) w1[Q] = tws[j * R1 + Q];
      });
      const int n0 = 2 * (j * (int)inner2 + c0 + t);          // window index of the pair's first sample (Q = 0)
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        int a = (int)(short)(raw[Q] & 0xffffu), b = (int)(short)(raw[Q] >> 16);
        if (p.derand) { a ^= -(a & 1) & 0xfffe; a = (int)(short)a; b ^= -(b & 1) & 0xfffe; b = (int)(short)b; }
        float fa = (float)a * p.scale16, fb = (float)b * p.scale16;   // rounded products, as convert() stores them
        CHZ_ROUNDED_F32(fa); CHZ_ROUNDED_F32(fb);
        v[Q] = make_float2(fa, fb);
        const int n = n0 + 2 * Q * qstep;
        if (n >= p.new_from) { energy += (unsigned)(a * a); clips += (a > 32766 || a < -32766); }
        if (n + 1 >= p.new_from) { energy += (unsigned)(b * b); clips += (b > 32766 || b < -32766); }
      });
    }
    reg_dft<R1, -1>(v);
    static_for<R1>([&](auto k1) {
      constexpr int K1 = decltype(k1)::value;
      float2 x = v[K1];
      if constexpr (K1 > 0) x = cmul(x, w1[K1]);
      lds[j * T + t + K1 * (R2 * T + p.padk)] = x;
    });
  }
  if (p.energy_part != nullptr) {     // workgroup-uniform; every lane takes part in the shuffles
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { energy += __shfl_xor(energy, d); clips += __shfl_xor(clips, d); }
    if ((tid & 63) == 0) {
      const int w = tile * ((nthr + 63) >> 6) + (tid >> 6);
      p.energy_part[w] = energy; p.clip_part[w] = clips;
    }
  }
  __syncthreads();
  // layer 2: radix R2 over j, one (k1, column) pair per thread
  float2 u[R2];
  const int k1 = tid / T, t2 = tid - k1 * T;
  const bool act2 = tid < R1 * T;
  if (act2) {
    static_for<R2>([&](auto j) {
      constexpr int J = decltype(j)::value;
      u[J] = lds[k1 * (R2 * T + p.padk) + t2 + J * T];
    });
    reg_dft<R2, -1>(u);
    // Z[k1 + R1*k2] goes to a SECOND LDS region in natural row order for the Hermitian split,
    // so no barrier is needed between the layer-2 reads above and these writes
    static_for<R2>([&](auto k2) {
      constexpr int K2 = decltype(k2)::value;
      zl[k1 * T + t2 + K2 * (R1 * T)] = u[K2];
    });
  }
  __syncthreads();
  // split + twiddle + store:  real column 2p   -> (Z[k] + conj Z[Na-k]) / 2
  //                           real column 2p+1 -> (Z[k] - conj Z[Na-k]) / 2i
  // (the factors 1/2 and 1/2i live in the column twiddle table)
  float4* __restrict__ gout4 = reinterpret_cast<float4*>(gout);
  CHZ_OUT_DESC(odesc, gout4);
  const int orow = p.inner >> 1;                   // row pitch of buf in float4 units
  static_for<NE>([&](auto uu) {
    constexpr int U = decltype(uu)::value;
    const int k = kr + U * rpi;
    if (lane_on && k < p.Ra) {
      const float2 a = zl[k * T + pc];
      const float2 b = zl[(k == 0 ? 0 : NA - k) * T + pc];
      const float2 de = make_float2(a.x + b.x, a.y - b.y);   // a + conj(b)
      const float2 dd = make_float2(a.x - b.x, a.y + b.y);   // a - conj(b)
      const float2 oe = cmul(de, cmul(we1[U], make_float2(we2[U].x, we2[U].y)));
      const float2 oo = cmul(dd, cmul(we1[U], make_float2(we2[U].z, we2[U].w)));
      CHZ_STORE(odesc, gout4, k * orow + c0 + pc, make_float4(oe.x, oe.y, oo.x, oo.y));
    }
  });
}

// ------------------------------------------------------------------------------
// K1b: column FFTs along a strided axis, T adjacent columns per workgroup.
// grid = rows * inner/T.
// ------------------------------------------------------------------------------
template <int R1, int R2>
__global__ void fwd_cols(ColsParams p) {
  constexpr int NP = R1 * R2;
  HIP_DYNAMIC_SHARED(float2, lds)
  const int tid = threadIdx.x;
  const int T = p.T;
  const int tpr = p.inner / T;
#if CHZ_XCD_AFFINE
  int row, ct;
  {
    const int comp = (((int)blockIdx.x & 7) + 8 - p.xa.rot) & 7, idx = (int)blockIdx.x >> 3;
    if (comp >= p.xa.ncomp) return;
    const int r0 = comp * p.xa.Ta - p.xa.shift;
    const int lo = r0 < 0 ? 0 : r0, hi = r0 + p.xa.Ta < p.rows ? r0 + p.xa.Ta : p.rows;
    if (idx >= (hi - lo) * tpr) return;
    row = lo + idx / tpr; ct = idx - (idx / tpr) * tpr;
  }
#else
  const int row = blockIdx.x / tpr, ct = blockIdx.x - row * tpr;
#endif
  const int c0 = ct * T;
  const long base = (long)row * NP * p.inner + c0;
  const float2* __restrict__ gin = CHZ_BSEL(p.bbuf, p.in);
  float2* __restrict__ gout = CHZ_BSEL(p.bbuf, p.out);
  const float2* __restrict__ tws = p.tw_sub;
  const float2* __restrict__ twt = p.tw_tile;
  const float2* __restrict__ twc = p.tw_col;

  // The output twiddles of the SECOND layer depend only on (k1, column): fetch them first so
  // their latency hides under the data loads and the first butterfly layer.
  const int k1o = tid / T, to = tid - k1o * T;
  const bool act2 = tid < R1 * T;
  float2 wt[R2], wc[R2];
  const bool full = p.tw_full != nullptr;                 // uniform
  if (act2) {
    if (full) {
      const float2* __restrict__ twf = p.tw_full + c0 + to;
      static_for<R2>([&](auto k2) { constexpr int K2 = decltype(k2)::value; wt[K2] = twf[(k1o + R1 * K2) * p.inner]; });
    } else {
      static_for<R2>([&](auto k2) {
        constexpr int K2 = decltype(k2)::value;
        const int k = k1o + R1 * K2;
        wt[K2] = twt[ct * NP + k];
        wc[K2] = twc[k * T + to];
      });
    }
  }
  const int gstep = R2 * T + p.padk;             // LDS distance between butterfly groups
  if (tid < R2 * T) {
    const int j = tid / T, t = tid - j * T;
    float2 v[R1], w1[R1];
    // 32-bit offsets from one 64-bit base; the ring wrap (complex masters) is a compare per row
    const int qstep = R2 * p.inner;
    if (p.in_len) {
      const int first = (int)((base + p.in_start + (long)j * p.inner + t) % p.in_len);
      const int len = (int)p.in_len;
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        int idx = first + Q * qstep;               // first < len and Q*qstep < N <= len: one wrap at most
        if (idx >= len) idx -= len;
        v[Q] = gin[idx];
        if constexpr (Q > 0) w1[Q] = tws[j * R1 + Q];
      });
    } else {
      const float2* __restrict__ g0 = gin + base + (long)j * p.inner + t;
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        v[Q] = g0[Q * qstep];
        if constexpr (Q > 0) w1[Q] = tws[j * R1 + Q];
      });
    }
    reg_dft<R1, -1>(v);
    const int l1 = j * T + t;
    static_for<R1>([&](auto k1) {
      constexpr int K1 = decltype(k1)::value;
      float2 x = v[K1];
      if constexpr (K1 > 0) x = cmul(x, w1[K1]);
      lds[l1 + K1 * gstep] = x;
    });
  }
  __syncthreads();
  if (act2) {
    float2 u[R2];
    const int l2 = k1o * gstep + to;
    static_for<R2>([&](auto j) {
      constexpr int J = decltype(j)::value;
      u[J] = lds[l2 + J * T];
    });
    reg_dft<R2, -1>(u);
    CHZ_OUT_DESC(odesc, gout);
    const int o0 = (int)base + k1o * p.inner + to;       // element index into gout: the whole buffer is < 2^31 bytes
    const int ostep = R1 * p.inner;
#if CHZ_XCD_AFFINE == 1        // plain stores: the lines stay in this XCD's L2 for fwd_rows (a COMPILE-time choice: an untaken run-time
    static_for<R2>([&](auto k2) {  // branch around a second store flavour cost this pass 6 us, 5.7 -> 12.0)
      constexpr int K2 = decltype(k2)::value;
      gout[o0 + K2 * ostep] = cmul(u[K2], full ? wt[K2] : cmul(wt[K2], wc[K2]));
    });
#else
    static_for<R2>([&](auto k2) {
      constexpr int K2 = decltype(k2)::value;
      CHZ_STORE(odesc, gout, o0 + K2 * ostep, cmul(u[K2], full ? wt[K2] : cmul(wt[K2], wc[K2])));
    });
#endif
  }
}

// K2 inside fwd_rows: the thread that is about to store a listed bin takes the block's ticket and runs the reference's recurrence
//   state += alpha * (X[bin] - state);  X[bin] -= state        (src/filter.c:464-474; state double complex, X float complex)
// over the head entry and every entry chained behind it (the same bin named again), in list order.  A wait that runs out, a
// tombstone left by an earlier failure, or a bin that is not the one the host named (a planner/owner mismatch: never seen, checked
// anyway) leave the value as it is, raise the host-visible error word and set the tombstone: a wrong recurrence is never published.
__device__ __forceinline__ float2 rows_notch_apply(const RowsNotch& nf, int e, int addr_e, double alpha_e, float2 x, int at, bool stored) {
  if (at != addr_e || !stored) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (nf.ver != nullptr) __hip_atomic_store(nf.ver + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nf.err != nullptr) __hip_atomic_store(nf.err, 0x80000000u | (unsigned)e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    if (nf.err != nullptr) *nf.err = 0x80000000u | (unsigned)e;
#endif
    return x;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (nf.ver != nullptr) {
    if (__hip_atomic_load(nf.ver + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return x;
    unsigned seq = nf.seq;
    if (nf.seq_base != nullptr) seq += __hip_atomic_load(nf.seq_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v = 0;
    const long long t0 = wall_clock64();
    for (;;) {
      v = __hip_atomic_load(nf.ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v == seq || wall_clock64() - t0 > nf.max_wait) break;
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (v != seq) {
      __hip_atomic_store(nf.ver + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(nf.err, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return x;
    }
  }
#endif
  for (int q = e; q >= 0; q = nf.next[q]) {
    double sr = nf.state[2 * q], si = nf.state[2 * q + 1];
    const double al = q == e ? alpha_e : nf.alpha[q];
    double dr = al * ((double)x.x - sr), di = al * ((double)x.y - si);   // rounded products, then the sums (no fma on x86-64)
    CHZ_ROUNDED_F64(dr); CHZ_ROUNDED_F64(di);
    sr += dr; si += di;
    nf.state[2 * q] = sr; nf.state[2 * q + 1] = si;
    x = make_float2((float)((double)x.x - sr), (float)((double)x.y - si));
  }
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the state stores have left this wave before the workgroup's barrier: the publisher's write-back covers them
#endif
  return x;
}
// ... and once every owner workgroup of the block is through, the last of them hands the ticket on (release: the states and the
// notched bins of this block are visible to whoever acquires seq + 1).  Called by ALL threads of an owner workgroup.
__device__ __forceinline__ void rows_notch_publish(const RowsNotch& nf, int tid) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (nf.ver == nullptr) return;
  __syncthreads();                                     // this workgroup's state and bin writes precede the release
  if (tid == 0) {
    bool last = true;
    if (nf.nwg > 1) {
      const unsigned c = __hip_atomic_fetch_add(nf.ver + 3, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      last = c == (unsigned)nf.nwg;
      if (last) __hip_atomic_store(nf.ver + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (last && __hip_atomic_load(nf.ver + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      unsigned seq = nf.seq;
      if (nf.seq_base != nullptr) seq += __hip_atomic_load(nf.seq_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (nf.adv != 0u) __hip_atomic_store(nf.seq_base, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // release, spelled out: the fence writes the XCD's dirty lines back (the states; the bins went out write-through), the explicit
      // wait keeps the ticket from overtaking that write-back (ROCm 7.2 drops the wait behind buffer_wbl2 when it believes this wave's
      // memory counter is empty -- it is, the stores were another wave's: MI355X_MICROARCH.md, "compiler hazard"), then a relaxed store
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(nf.ver, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#else
  (void)nf; (void)tid;
#endif
}

// ------------------------------------------------------------------------------
// K1c: last axis.  grid = Nb * ceil(Ra/Ta); tile = Ta consecutive ka at one kb.
// ------------------------------------------------------------------------------
template <int R1, int R2>
__global__ void fwd_rows(RowsParams p) {
  constexpr int NC = R1 * R2;
  HIP_DYNAMIC_SHARED(float2, lds)
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int Ta = p.Ta, ld = p.ld, padg = p.padg;
#if CHZ_XCD_AFFINE
  int kb, at;
  {
    const int comp = (((int)blockIdx.x & 7) + 8 - p.xa.rot) & 7, idx = (int)blockIdx.x >> 3;
    if (comp >= p.xa.ncomp || idx >= p.Nb) return;
    kb = idx; at = comp;
  }
#else
  const int kb = blockIdx.x % p.Nb, at = blockIdx.x / p.Nb;
#endif
  const int a0 = at * Ta - p.ka_shift;           // may be negative for the first (ragged) tile
  const int rowstride = p.Nb * NC;               // all index math below is 32-bit: N < 2^31
  const int gstep = R2 * ld + padg;              // LDS distance between butterfly groups

  const float2* __restrict__ tws = p.tw_sub;
  const float2* __restrict__ gin = CHZ_BSEL(p.bbuf, p.buf) + (long)kb * NC;
  bool nf_owner = false;                           // does this workgroup store a bin of the notch list?  (scalar compares on kernel arguments)
  static_for<CHZ_NOTCH_INLINE>([&](auto ee) { constexpr int E = decltype(ee)::value; nf_owner = nf_owner || (E < p.nf.n && p.nf.wg[E] == (int)blockIdx.x); });
  unsigned nf_mask = 0;                            // ... and which of this thread's second-layer outputs, if any (at most one: rows_notch_fill)
  float2 nf_x = make_float2(0.f, 0.f); int nf_at = -1; bool nf_ok = false;
  int nf_e = -1, nf_addr = -1; double nf_alpha = 0.0;     // the entry, fetched now: only the ticket and the state are left for the tail
  if (nf_owner)
    for (int e = 0; e < p.nf.n; e++)
      if (p.nf.head[e] != 0 && p.nf.own[e].wg == (int)blockIdx.x && p.nf.own[e].tid == tid) {
        nf_mask |= 1u << p.nf.own[e].k2; nf_e = e; nf_addr = p.nf.addr[e]; nf_alpha = p.nf.alpha[e];
      }
  if constexpr (R2 % 16 == 0) {
    // The first layer takes its points straight from global memory: with the lanes running along nc (j1 fastest),
    // the R2 lanes of one row read R2 consecutive complex values = whole 128-byte lines per load instruction, and
    // the separate transposition pass through LDS (one barrier, one LDS write + read, its index arithmetic) is gone.
    const int r1 = tid / R2, j1 = tid - r1 * R2;
    const bool act1 = tid < R2 * Ta;
    if (act1) {
      const int ka = a0 + r1;
      const bool in = ka >= 0 && ka < p.Ra;
      const float2* __restrict__ g = gin + (in ? ka : 0) * rowstride + j1;
      float2 v[R1], w1[R1];
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        v[Q] = g[Q * R2];
        if constexpr (Q > 0) w1[Q] = tws[j1 * R1 + Q];
      });
      if (!in) static_for<R1>([&](auto q) { constexpr int Q = decltype(q)::value; v[Q] = make_float2(0.f, 0.f); });
      reg_dft<R1, -1>(v);
      const int b1 = j1 * ld + r1;                 // row K1*R2 + j1 (group K1), column r1
      static_for<R1>([&](auto k1) {
        constexpr int K1 = decltype(k1)::value;
        float2 x = v[K1];
        if constexpr (K1 > 0) x = cmul(x, w1[K1]);
        lds[b1 + K1 * gstep] = x;
      });
    }
  } else {
  // first-layer twiddles depend only on the lane: fetch them before anything else
  const int j1 = tid / Ta, r1 = tid - j1 * Ta;
  const bool act1 = tid < R2 * Ta;
  float2 w1[R1];
  static_for<R1>([&](auto q) {
    constexpr int Q = decltype(q)::value;
    if constexpr (Q > 0) w1[Q] = tws[(act1 ? j1 : 0) * R1 + Q];
  });
  // coalesced row loads, transposed into LDS as [nc][r]; LOAD_U loads are in flight per lane
  constexpr int LOAD_U = 6;
  for (int e0 = tid; e0 < Ta * NC; e0 += nthr * LOAD_U) {
    float2 x[LOAD_U];
    int la[LOAD_U];
    static_for<LOAD_U>([&](auto u) {
      constexpr int U = decltype(u)::value;
      const int e = e0 + U * nthr;
      const int r = e / NC, nc = e - r * NC;       // NC is a compile-time constant
      const int ka = a0 + r;
      la[U] = (e < Ta * NC) ? nc * ld + (nc / R2) * padg + r : -1;
      x[U] = make_float2(0.f, 0.f);
      if (la[U] >= 0 && ka >= 0 && ka < p.Ra) x[U] = gin[ka * rowstride + nc];
    });
    static_for<LOAD_U>([&](auto u) {
      constexpr int U = decltype(u)::value;
      if (la[U] >= 0) lds[la[U]] = x[U];
    });
  }
  __syncthreads();
  if (act1) {
    const int b1 = j1 * ld + r1;                   // row j1 + Q*R2 sits at b1 + Q*gstep
    float2 v[R1];
    static_for<R1>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      v[Q] = lds[b1 + Q * gstep];
    });
    reg_dft<R1, -1>(v);
    static_for<R1>([&](auto k1) {
      constexpr int K1 = decltype(k1)::value;
      float2 x = v[K1];
      if constexpr (K1 > 0) x = cmul(x, w1[K1]);
      lds[b1 + K1 * gstep] = x;                    // in place: row K1*R2 + j1 (group K1)
    });
  }
  }
  __syncthreads();
  if (tid < R1 * Ta) {
    const int k1 = tid / Ta, r = tid - k1 * Ta;
    const int ka = a0 + r;
    const int b2 = k1 * gstep + r;
    float2 u[R2];
    static_for<R2>([&](auto j) {
      constexpr int J = decltype(j)::value;
      u[J] = lds[b2 + J * ld];
    });
    reg_dft<R2, -1>(u);
    if (ka >= 0 && ka < p.Ra) {
      const bool selfconj = (ka == 0) || (2 * ka == p.Na);
      const int half = (int)(p.N >> 1);
      const int xrows = (int)(p.N / p.Na);         // = Nb*Nc
      const int x0 = kb + p.Nb * k1, xs = p.Nb * R1;              // x = x0 + K2*xs
      const int kk0 = ka + p.Na * x0, kks = p.Na * xs;            // bin index k = kk0 + K2*kks
      const int d0 = x0 * p.lay.pitch + p.lay.off + ka, ds = xs * p.lay.pitch;
      const int m0 = (xrows - 1 - x0) * p.lay.pitch + p.lay.off + (p.Na - ka);
      float2* __restrict__ sp = CHZ_BSEL(p.bspec, p.spec);
      CHZ_OUT_DESC(sdesc, sp);
      // one store per output, no branches: bins up to N/2 go out as they are, the rest conjugated to bin N-k; the
      // self-conjugate rows (ka = 0, Na/2) have no mirror image to write
      static_for<R2>([&](auto k2) {
        constexpr int K2 = decltype(k2)::value;
        const bool direct = !p.mirror || kk0 + K2 * kks <= half;
        const int at = direct ? d0 + K2 * ds : m0 - K2 * ds;
        float2 x = u[K2];
        if (!direct) x.y = -x.y;
        if (nf_mask >> K2 & 1u) { nf_x = x; nf_at = at; nf_ok = direct || !selfconj; }   // a listed bin: remember what goes where (one thread of the pass, as a rule)
        CHZ_STORE_IF(sdesc, sp, at, x, direct || !selfconj);
      });
      // K2: the listed bin is stored once more, notched (same lane, same address, program order; nobody reads the slot before the
      // kernel ends).  Done here, behind the store loop, so that the rare path does not cost the common one any registers.
      if (nf_mask != 0u) {
        const float2 y = rows_notch_apply(p.nf, nf_e, nf_addr, nf_alpha, nf_x, nf_at, nf_ok);
        CHZ_STORE_IF(sdesc, sp, nf_at, y, nf_ok);
      }
    }
  }
  if (nf_owner) rows_notch_publish(p.nf, tid);
}

// ------------------------------------------------------------------------------
// K2: spur notches (apply_notch_filters, src/filter.c:464-474): for every list entry, in list order,
//   state += alpha * (X[bin] - state);  X[bin] -= state          (state is double complex, X float complex)
// The state is a recurrence over BLOCKS, and blocks of different HIP streams run concurrently, so this tiny
// kernel (one workgroup) sits on the block's own stream right after fwd_rows and takes its turn in block order:
//   ver != nullptr   a ticket: wait until `ver` (blocks applied so far) equals this block's sequence number,
//                    apply, publish seq+1 with release semantics.  The engine issues these kernels in block order
//                    (chz_engine.hip: NotchTurn), so whatever a ticket waits for was enqueued BEFORE it on every
//                    hardware queue and cannot be stuck behind it; the wait is bounded all the same, and a wait
//                    that runs out touches nothing, raises `err` (host-visible) and leaves `ver` alone, so every
//                    later block fails too and the host reports it -- a wrong recurrence is never published.
//   ver == nullptr   ordering is done by the caller (HIP events between the streams, graph capture, one stream).
// One lane per distinct bin; entries that name the same bin again are chained through `next` and applied by the
// same lane in list order, exactly as the reference's sequential walk does.  Lists of up to CHZ_NOTCH_INLINE
// entries (radiod's usual DC-only or few-spur lists) travel in the kernel arguments: no dependent table loads.
// ------------------------------------------------------------------------------
struct NotchFixParams {
  float2* spec;           // this block's spectrum slot (SpecLayout order)
  const int* addr;        // [n] storage index of the entry's bin
  const int* next;        // [n] next entry with the same bin, or -1
  const int* head;        // [n] 1 if the entry is the first one naming its bin
  const double* alpha;    // [n] per-entry averager gain
  double* state;          // [n][2] persists across blocks
  int n;
  int inl;                // 1: the list is in the i_* arrays below
  unsigned* ver;          // [4] ticket counter, tombstone, graph base, spare; or nullptr
  unsigned seq;           // this block's ticket (relative to *seq_base when that is set)
  unsigned* seq_base;     // captured launches: the ticket is *seq_base + seq, so one captured kernel node serves every replay
  unsigned adv;           // != 0: this is the last block of the captured sequence; it moves *seq_base on to its own ticket + 1
  unsigned* err;          // host-visible error word (0 = fine)
  long long max_wait;     // ticket wait budget in ticks of the constant-rate counter (hipDeviceAttributeWallClockRate)
  int i_addr[CHZ_NOTCH_INLINE];
  signed char i_next[CHZ_NOTCH_INLINE], i_head[CHZ_NOTCH_INLINE];
  double i_alpha[CHZ_NOTCH_INLINE];
};
__global__ void __launch_bounds__(1024) notch_fix(NotchFixParams p) {
  const int i = (int)threadIdx.x;
  // loads that do not depend on the previous block go out before the wait
  bool mine = false; int a = 0;
  if (i < p.n) {
    if (p.inl) { mine = p.i_head[i] != 0; a = p.i_addr[i]; }
    else { mine = p.head[i] != 0; a = p.addr[i]; }
  }
  float2 x = mine ? p.spec[a] : make_float2(0.f, 0.f);
  unsigned seq = p.seq; (void)seq;
#if defined(__HIP_DEVICE_COMPILE__)
  if (p.ver != nullptr) {
    // poll with relaxed loads (they bypass the non-coherent caches); ONE acquire fence once the ticket is up, so the
    // cache invalidation that comes with it is not repeated per poll
    // bounded in TIME (constant-rate counter), not in polls: free-running over banks of millions of channels a predecessor
    // can legitimately be tens of milliseconds away, queued behind its lane's previous channel kernels
    // p.ver[1] is the chain's tombstone: once one wait has run out nobody behind it waits again (they would each sit out the
    // whole budget: the counter never moves on after a failure)
    if (__hip_atomic_load(p.ver + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    // A captured launch cannot carry a running number in its arguments: its ticket is relative to a base word that the
    // last block of each replay moves on.  Every other block of the replay has read the base before the last one's
    // turn comes (their turns precede it), and the next replay starts after this one has finished (stream order).
    if (p.seq_base != nullptr) seq += __hip_atomic_load(p.seq_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned v = 0;
    const long long t0 = wall_clock64();
    for (;;) {
      v = __hip_atomic_load(p.ver, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v == seq || wall_clock64() - t0 > p.max_wait) break;
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (v != seq) {                                     // never publish a wrong recurrence
      if (i == 0) {
        __hip_atomic_store(p.ver + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.err, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
#endif
  if (mine) {
    for (int e = i; e >= 0; e = p.inl ? (int)p.i_next[e] : p.next[e]) {
      double sr = p.state[2 * e], si = p.state[2 * e + 1];
      const double al = p.inl ? p.i_alpha[e] : p.alpha[e];
      double dr = al * ((double)x.x - sr), di = al * ((double)x.y - si);   // rounded products, then the sums (no fma on x86-64)
      CHZ_ROUNDED_F64(dr); CHZ_ROUNDED_F64(di);
      sr += dr; si += di;
      p.state[2 * e] = sr; p.state[2 * e + 1] = si;
      x = make_float2((float)((double)x.x - sr), (float)((double)x.y - si));
    }
    p.spec[a] = x;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (p.ver != nullptr) {
    __syncthreads();                                     // every lane's state and bin writes precede the release
    if (i == 0) {
      if (p.adv != 0u) __hip_atomic_store(p.seq_base, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the last block's ticket + 1 = base + blocks per replay
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // (spelled out as in rows_notch_publish: the wait must not be optimised away)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(p.ver, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#endif
}

// ------------------------------------------------------------------------------
// K3+K4: per-channel gather x response, P-point backward FFT, keep last olen.
// LPC = max(R1,R2) lanes serve one channel, 64/LPC channels share a wavefront.
// ------------------------------------------------------------------------------
// EPI: 0 = plain execute_filter_output; 1 = with the downconvert() epilogue (fine tuning, power, the AGC's slice-energy peak); 2 = that plus the
// ISB unpack and beam mode.  Three instantiations because the rare paths cost the common ones registers: the full variant of the 12 kHz
// kernel takes 150 VGPRs (3 wavefronts per SIMD) -- beam mode's double-precision products over R1 registers, the mirror bins, the ISB
// partner values -- which every TUNED bank paid for although hardly any uses them (round 4).
// (the small sizes are asked to fit 6 wavefronts per SIMD: without the hint the 12 kHz kernel takes 95 VGPRs and loses one)
template <int R1, int R2, int EPI>
__global__ void __launch_bounds__(256, (R1 <= 15 && R2 <= 20) ? (EPI == 0 ? 6 : (EPI == 1 ? 5 : 1)) : 1) chan_ifft(ChanParams p) {
  constexpr int P = R1 * R2;
  constexpr int LPC = R1 > R2 ? R1 : R2;
  constexpr int CPW = 64 / LPC;
  constexpr int LDC = R2 + 1;                    // padded row of the exchange buffer
  static_assert(CPW >= 1, "radix too wide for one wavefront");
  HIP_DYNAMIC_SHARED(float2, lds)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wpb = blockDim.x >> 6;
  const int cw = lane / LPC, jl = lane - cw * LPC;
  const int lc = (blockIdx.x * wpb + wave) * CPW + cw;
  const int ch = p.ch0 + lc;
  const bool live = (cw < CPW) && (lc < p.nch);
  float2* my = lds + ((wave * CPW + (cw < CPW ? cw : 0)) * (R1 * LDC));
  const float2* __restrict__ tws = p.tw_sub;

  float2 v[R1];
  const bool act1 = live && jl < R2;
  if (act1) {
    const ChanDesc d = p.desc[ch];
    const float2* __restrict__ H = p.resp + (long)d.row * P;
    const float2* __restrict__ X = p.spec;
    // All 2*R1 loads are issued unconditionally (out-of-range bins read bin 0 and are
    // zeroed afterwards) so they overlap instead of costing one round trip per bin.
    float2 h[R1];
    bool ok[R1];
    int srcs[EPI == 2 ? R1 : 1];                          // master bin of each register (beam mode needs its mirror)
    // Index arithmetic is what this kernel spends most of its instructions on (at millions of channels it is bound by them, not by
    // memory): the +-1 direction is a sign trick, not a multiply; both range tests are one unsigned compare; bin -> storage index
    // is a multiply-high with a host-made reciprocal instead of a float estimate with fix-ups; the gather goes through a buffer
    // descriptor (32-bit offsets from a wave-uniform base) and the response row through one per-lane pointer with immediates.
    CHZ_IN_DESC(xdesc, X);
    const int dm = d.dir >> 31;                            // 0 for +1, -1 for -1
    const float2* __restrict__ Hl = H + jl;
    static_for<R1>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      constexpr int H2 = (P + 1) / 2;
      const int i = jl + Q * R2;                           // FFT-order bin index
      int t = i - H2; if (t < 0) t += P;                   // rank from most negative bin
      const int u = t - d.t0;
      ok[Q] = ((unsigned)u < (unsigned)d.cnt) && (i != H2);    // 0 <= u < cnt; Nyquist bin forced to zero (:911)
      int src = d.src0 + ((u ^ dm) - dm);                  // src0 + dir * u
      if (d.wrap && src >= d.wrap) src -= d.wrap;
      if (!ok[Q]) src = 0;
      v[Q] = CHZ_LOAD2(xdesc, X, spec_index(p.lay.off, p.magic, p.dpitch, src));
      h[Q] = CHZ_LOAD_RESP(Hl + Q * R2);
      if constexpr (EPI == 2) srcs[Q] = src;
    });
    bool beamed = false;
    if constexpr (EPI == 2) {
      if (p.beam != nullptr) {                             // wave-uniform
        const BeamDesc bd = p.beam[ch];
        if (bd.on && d.wrap > 0) {                         // COMPLEX masters only (wrap = master bins)
          beamed = true;
          float2 w[R1];
          static_for<R1>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            int mp = srcs[Q] == 0 ? 0 : d.wrap - srcs[Q];  // the mirror bin
            int row = (int)((float)mp * p.inv_na), col = mp - row * p.lay.na;
            if (col < 0) { row--; col += p.lay.na; } else if (col >= p.lay.na) { row++; col -= p.lay.na; }
            w[Q] = X[(long)row * p.lay.pitch + p.lay.off + col];
          });
          static_for<R1>([&](auto q) {
            constexpr int Q = decltype(q)::value;
            const double xr = v[Q].x, xi = v[Q].y, hr = h[Q].x, hi = h[Q].y;
            double sr, si;
            if (srcs[Q] == 0 || srcs[Q] == d.wrap / 2) {   // :766-768: Re X alpha H + Im X beta H
              const double t1r = xr * bd.ar, t1i = xr * bd.ai, t2r = xi * bd.br, t2i = xi * bd.bi;
              sr = (t1r * hr - t1i * hi) + (t2r * hr - t2i * hi);
              si = (t1r * hi + t1i * hr) + (t2r * hi + t2i * hr);
            } else {                                       // :770-771
              const double yr = w[Q].x, yi = -(double)w[Q].y;
              const double cr = (bd.ar * xr - bd.ai * xi) + (bd.br * yr - bd.bi * yi);
              const double ci = (bd.ar * xi + bd.ai * xr) + (bd.br * yi + bd.bi * yr);
              sr = cr * hr - ci * hi; si = cr * hi + ci * hr;
            }
            v[Q] = ok[Q] ? make_float2((float)sr, (float)si) : make_float2(0.f, 0.f);
          });
        }
      }
    }
    if (!beamed) {
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        float2 x = v[Q];
        if (d.conj) x.y = -x.y;
        x = cmul(x, h[Q]);
        v[Q] = ok[Q] ? x : make_float2(0.f, 0.f);
      });
    }
  }
  if constexpr (EPI == 2) {
    // ISB mode (slave->isb, src/filter.c:895-909, filter2 of the linear demodulator): LSB and USB are unpacked to
    // I and Q -- Y[p] += conj(Y[P-p]), Y[P-p] -= conj(Y[p]) for 0 < p < P/2, Y[0] = 0.  Bin P-i of lane jl, register Q
    // sits in lane R2-jl, register R1-1-Q (lane 0: its own register R1-Q): one cross-lane fetch per register.
    if (p.isb != nullptr) {                                // wave-uniform
      const bool mine = act1 && p.isb[ch] != 0;
      const int src_lane = (cw < CPW ? cw : 0) * LPC + (jl == 0 || jl >= R2 ? 0 : R2 - jl);
      float2 part2[R1];
      static_for<R1>([&](auto q) {
        constexpr int Q = decltype(q)::value;
        const float2 mine_v = act1 ? v[R1 - 1 - Q] : make_float2(0.f, 0.f);
        part2[Q] = make_float2(__shfl(mine_v.x, src_lane), __shfl(mine_v.y, src_lane));
      });
      if (mine) {
        float2 orig[R1];
        static_for<R1>([&](auto q) { constexpr int Q = decltype(q)::value; orig[Q] = v[Q]; });
        static_for<R1>([&](auto q) {
          constexpr int Q = decltype(q)::value;
          const int i = jl + Q * R2;
          float2 partner = part2[Q];                       // Y[P-i] for jl >= 1
          if (jl == 0) { if constexpr (Q >= 1) partner = orig[R1 - Q]; else partner = make_float2(0.f, 0.f); }
          if (i == 0) v[Q] = make_float2(0.f, 0.f);
          else if (2 * i < P) v[Q] = make_float2(orig[Q].x + partner.x, orig[Q].y - partner.y);        // pos + conj(neg)
          else if (2 * i > P) v[Q] = make_float2(orig[Q].x - partner.x, orig[Q].y + partner.y);        // neg - conj(pos)
          if (i == (P + 1) / 2) v[Q] = make_float2(0.f, 0.f);                                          // :911 comes after the unpack
        });
      }
    }
  }
  if (act1) {
    reg_dft<R1, +1>(v);
    static_for<R1>([&](auto k1) {
      constexpr int K1 = decltype(k1)::value;
      float2 x = v[K1];
      if constexpr (K1 > 0) x = cmul(x, tws[jl * R1 + K1]);
      my[K1 * LDC + jl] = x;
    });
  }
  __syncthreads();
  double part = 0.0;                                       // this lane's share of the block energy
  float2 u[R2];
  if (live && jl < R1) {
    static_for<R2>([&](auto j) {
      constexpr int J = decltype(j)::value;
      u[J] = my[jl * LDC + J];
    });
  }
  CHZ_WAVE_SYNC();                                         // the exchange buffer is reused for the output rows below
  if (live && jl < R1) {
    reg_dft<R2, +1>(u);
    const int drop = P - p.olen;                           // first M-1 samples are discarded (:357)
    if (EPI && p.fine != nullptr) {
      const FineDesc f = p.fine[ch];
      if (f.on) {
        // phase (cycles) of output sample m = jl - drop of this block; later samples step by R1
        const double kb = (double)(p.job - f.job0);
        const int m0 = jl - drop;
        const double g0 = kb * (double)p.olen + (double)m0;
        // r = ((kb mod V) + 1) * adj_num mod V, the phase_adjust multiplications so far (src/radio.c:1493,1497), without an integer
        // division: kb mod V from the two residues the host took (the block counter wraps at 2^32), the product's residue in double
        // (exact: V < 2^26, chz_bank_set_tuning)
        const int V = p.fine_V;
        int kbm = p.fine_jobm - f.job0m + (p.job < f.job0 ? p.fine_wrapm : 0);
        kbm += kbm < 0 ? V : 0; kbm -= kbm >= V ? V : 0;
        const double x = (double)(kbm + 1) * (double)f.adj_num;
        double r = fma(-floor(x * p.fine_invV), (double)V, x);
        r += r < 0.0 ? (double)V : 0.0; r -= r >= (double)V ? (double)V : 0.0;
        const double base = f.phase0 + r * p.fine_invV;
        if (f.rate == 0.0) {
          double hi = g0 * f.freq, lo = fma(g0, f.freq, -hi);          // exact product, reduced mod 1
          hi -= rint(hi);
          double s0, c0, s1, c1;
          sincospi(2.0 * (base + hi + lo), &s0, &c0);
          c1 = f.step_c; s1 = f.step_s;                                  // e^{2 pi i R1 freq}, the host's (fine_desc(.., stride = R1))
          static_for<R2>([&](auto k2) {
            constexpr int K2 = decltype(k2)::value;
            const double xr = u[K2].x, xi = u[K2].y;
            u[K2] = make_float2((float)(xr * c0 - xi * s0), (float)(xr * s0 + xi * c0));
            const double nc = c0 * c1 - s0 * s1; s0 = c0 * s1 + s0 * c1; c0 = nc;
          });
        } else {
          static_for<R2>([&](auto k2) {
            constexpr int K2 = decltype(k2)::value;
            const double g = g0 + (double)(R1 * K2);
            double hi = g * f.freq, lo = fma(g, f.freq, -hi);
            hi -= rint(hi);
            double q = 0.5 * f.rate * g * (g + 1.0);
            q -= rint(q);
            double sn, cs;
            sincospi(2.0 * (base + hi + lo + q), &sn, &cs);
            const double xr = u[K2].x, xi = u[K2].y;
            u[K2] = make_float2((float)(xr * cs - xi * sn), (float)(xr * sn + xi * cs));
          });
        }
      }
    }
    if (p.stage) {
      static_for<R2>([&](auto k2) {
        constexpr int K2 = decltype(k2)::value;
        const int n = jl + R1 * K2;
        if (n >= drop) my[n - drop] = u[K2];
      });
    } else {
      float2* __restrict__ o = p.out + (long)ch * p.olen;
      static_for<R2>([&](auto k2) {
        constexpr int K2 = decltype(k2)::value;
        const int n = jl + R1 * K2;
        if (n >= drop) o[n - drop] = u[K2];
      });
    }
    if constexpr (EPI) {
      static_for<R2>([&](auto k2) {
        constexpr int K2 = decltype(k2)::value;
        if (jl + R1 * K2 >= drop) part += (double)(u[K2].x * u[K2].x + u[K2].y * u[K2].y);
      });
    }
  }
  CHZ_WAVE_SYNC();
  // The channels of one wavefront are neighbours in the bank, so their olen-sample rows are ONE contiguous run of
  // global memory: stream it out of LDS as 16-byte stores in lane order (full 128-byte lines) instead of the
  // R1-sample pieces the butterfly leaves in each lane.  Worth 11 % at millions of channels (12 M channels: 12.8 ->
  // 11.4 ms per block), costs 0.6 us of latency on a 1024-channel launch: the engine picks per launch.
  if constexpr (EPI) {
    if (p.stage && p.agc_peak != nullptr) {                 // wave-uniform
      // slice s = samples [s*sps, (s+1)*sps) of the channel's row in LDS; the channel's lanes take the slices in turn and sum each one in
      // sample order exactly as demod_lin_lanes' first pass did (float squares rounded one by one, their float sum, accumulated in double);
      // a slice counts if it ends before the block's last sample (`while (n + samples_per_slice < N)`, src/linear.c:199)
      const int sps = p.agc_sps, N = p.olen;
      double peak = 0.0;
      if (live) {
        for (int s0 = jl; (s0 + 1) * sps < N; s0 += LPC) {
          double energy = 0.0;
          for (int n = s0 * sps; n < (s0 + 1) * sps; n++) {
            const float2 w = my[n];
            float a = w.x * w.x, b = w.y * w.y;
            CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
            energy += (double)(a + b);
          }
          if (energy > peak) peak = energy;
        }
      }
      // max over the channel's lanes (energies are >= 0: max of the lanes' maxima), a tree over the group: lane jl takes lane jl + d
      double tot = peak;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) if (d < LPC) { const double o = __shfl_down(tot, (unsigned)d); if (jl + d < LPC && o > tot) tot = o; }
      if (live && jl == 0) p.agc_peak[ch] = tot;
    }
  }
  if (p.stage) {
    const int first_lc = (blockIdx.x * wpb + wave) * CPW;
    int nl = p.nch - first_lc; if (nl > CPW) nl = CPW;
    const int tot2 = nl > 0 ? (nl * p.olen) >> 1 : 0;      // float4 = two samples; olen is even (P = olen*N/L, :312)
    const float2* wl = lds + (wave * CPW) * (R1 * LDC);
    float4* __restrict__ og = reinterpret_cast<float4*>(p.out + (long)(p.ch0 + first_lc) * p.olen);
    (void)tot2;
    const int half = p.olen >> 1;                          // float4 = two samples; olen is even (P = olen*N/L, :312)
    for (int c = 0; c < nl; c++) {                         // channel by channel: no division by a run-time olen per element
      const float2* wc = wl + c * (R1 * LDC);
      float4* __restrict__ oc = og + (long)c * half;
      for (int e = lane; e < half; e += 64) {
        const float2 a = wc[2 * e], b = wc[2 * e + 1];
        CHZ_STORE_OUT4(oc + e, make_float4(a.x, a.y, b.x, b.y));
      }
    }
  }
  if (EPI && p.power != nullptr) {     // wave-uniform: every lane takes part in the shuffles
    double tot = part;                                       // (lanes outside the second layer hold 0)
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) if (d < LPC) { const double o = __shfl_down(tot, (unsigned)d); if (jl + d < LPC) tot += o; }
    if (live && jl == 0) p.power[ch] = tot / (double)p.olen;
  }
}

// ------------------------------------------------------------------------------
// K3+K4 for REAL-output slaves (create_filter_output(.., REAL): wfm's composite filters): the gather of
// src/filter.c:803-809 (REAL master) / :794-802 (COMPLEX master) fills bins 0..P/2, bin (bins+1)/2 is zeroed
// (:911), and the c2r transform of src/filter.c:914 via :387 is a backward transform of the Hermitian
// extension Y[P-k] = conj(Y[k]) with the imaginary parts of DC and Nyquist ignored, as FFTW's c2r does.
// Output: the last olen of P real samples (:385).  Same lane layout as chan_ifft; P even.
// ------------------------------------------------------------------------------
template <int R1, int R2>
__global__ void __launch_bounds__(256) chan_c2r(ChanParams p) {
  constexpr int P = R1 * R2;
  constexpr int LPC = R1 > R2 ? R1 : R2;
  constexpr int CPW = 64 / LPC;
  constexpr int LDC = R2 + 1;
  static_assert(CPW >= 1, "radix too wide for one wavefront");
  static_assert(P % 2 == 0, "real-output channels need an even P");
  HIP_DYNAMIC_SHARED(float2, lds)
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wpb = blockDim.x >> 6;
  const int cw = lane / LPC, jl = lane - cw * LPC;
  const int lc = (blockIdx.x * wpb + wave) * CPW + cw;
  const int ch = p.ch0 + lc;
  const bool live = (cw < CPW) && (lc < p.nch);
  float2* my = lds + ((wave * CPW + (cw < CPW ? cw : 0)) * (R1 * LDC));
  const float2* __restrict__ tws = p.tw_sub;

  if (live && jl < R2) {
    const ChanDesc d = p.desc[ch];
    const int shift = d.shift;
    const float2* __restrict__ H = p.resp + (long)d.row * P;
    const float2* __restrict__ X = p.spec;
    constexpr int SB = P / 2 + 1;                          // slave bins (:374)
    float2 v[R1], w[R1], h[R1];
    bool ok[R1];
    static_for<R1>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      const int i = jl + Q * R2;                           // index into the Hermitian-extended spectrum
      const int k = i <= P / 2 ? i : P - i;                // the slave bin it comes from
      const int mi = k + shift;
      int a, b = 0;
      if (p.m_real) {
        ok[Q] = mi >= 0 && mi < p.m_bins;                                           // :808
        a = ok[Q] ? mi : 0;
      } else {
        ok[Q] = mi >= -(p.m_bins / 2) && mi < p.m_bins / 2;                         // :798
        a = mi % p.m_bins; if (a < 0) a += p.m_bins;
        b = (p.m_bins - mi) % p.m_bins; if (b < 0) b += p.m_bins;
        if (!ok[Q]) { a = 0; b = 0; }
      }
      if (k == (SB + 1) / 2) ok[Q] = false;                                         // :911
      int row = (int)((float)a * p.inv_na), col = a - row * p.lay.na;
      if (col < 0) { row--; col += p.lay.na; } else if (col >= p.lay.na) { row++; col -= p.lay.na; }
      v[Q] = X[(long)row * p.lay.pitch + p.lay.off + col];
      if (!p.m_real) {
        int rowb = (int)((float)b * p.inv_na), colb = b - rowb * p.lay.na;
        if (colb < 0) { rowb--; colb += p.lay.na; } else if (colb >= p.lay.na) { rowb++; colb -= p.lay.na; }
        w[Q] = X[(long)rowb * p.lay.pitch + p.lay.off + colb];
      }
      h[Q] = H[k];
    });
    static_for<R1>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      const int i = jl + Q * R2;
      const int k = i <= P / 2 ? i : P - i;
      float2 x = v[Q];
      if (!p.m_real) x = make_float2(x.x + w[Q].x, x.y - w[Q].y);                   // X[a] + conj(X[b]), :800
      x = cmul(x, h[Q]);
      if (k == 0 || 2 * k == P) x.y = 0.f;                                          // c2r ignores these imaginary parts
      if (i > P / 2) x.y = -x.y;                                                    // Hermitian extension
      v[Q] = ok[Q] ? x : make_float2(0.f, 0.f);
    });
    reg_dft<R1, +1>(v);
    static_for<R1>([&](auto k1) {
      constexpr int K1 = decltype(k1)::value;
      float2 x = v[K1];
      if constexpr (K1 > 0) x = cmul(x, tws[jl * R1 + K1]);
      my[K1 * LDC + jl] = x;
    });
  }
  __syncthreads();
  if (live && jl < R1) {
    float2 u[R2];
    static_for<R2>([&](auto j) {
      constexpr int J = decltype(j)::value;
      u[J] = my[jl * LDC + J];
    });
    reg_dft<R2, +1>(u);
    const int drop = P - p.olen;
    float* __restrict__ o = reinterpret_cast<float*>(p.out) + (long)ch * p.olen;
    static_for<R2>([&](auto k2) {
      constexpr int K2 = decltype(k2)::value;
      const int n = jl + R1 * K2;
      if (n >= drop) o[n - drop] = u[K2].x;
    });
  }
}

// ------------------------------------------------------------------------------
// Masters of ANY length (FFTW plans every N, src/filter.c:101-163,222-231; the compiled axes cover N = a x b x c from a menu).
// Bluestein's identity  n k = (n^2 + k^2 - (k - n)^2) / 2  turns the N-point transform into a circular convolution of length
// Mz >= 2N - 1, Mz a planned size:   X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k - n],   c[n] = exp(-i pi n^2 / N).
// Three elementwise kernels around two runs of the complex forward transform of length Mz (the inverse as conj(F(conj(.))) / Mz):
//   blue_pre   za[n] = x[n] c[n] (window read out of the sample ring), zero beyond N
//   blue_mul   za[k] = conj(Z[k] * Bf[k])            Z = F(za) in the planned transform's own storage order, Bf = F(conj(c) wrapped)
//   blue_post  X[k]  = c[k] * conj(Z'[k]) / Mz       the master's bins, in the master's spectrum layout
// ------------------------------------------------------------------------------
struct BluePreParams {
  const float* ring; long ring_len, start; int per; const float2* chirp; float2* za; int N; long Mz;
  // SURVEY 8(f) rank 3 on a chirp-z master (round 4): raw A/D samples converted on load, as fwd_first_real does (src/rx888.c:753-767)
  const short* ring16; float scale16; int derand, new_from;
  unsigned long long* energy_part; unsigned* clip_part;              // [grid * waves] partials over the block's NEW samples, or nullptr
};
__global__ void __launch_bounds__(256) blue_pre(BluePreParams p) {
  const long stride = (long)gridDim.x * blockDim.x;
  unsigned long long energy = 0; unsigned clips = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.Mz; i += stride) {
    float2 v = make_float2(0.f, 0.f);
    if (i < p.N) {
      long idx = p.start + i * p.per;
      if (idx >= p.ring_len) idx -= p.ring_len;                      // (a window is never longer than the ring)
      float2 x;
      if (p.ring16 != nullptr) {                                     // REAL masters only (per == 1)
        int a = (int)p.ring16[idx];
        if (p.derand) { a ^= -(a & 1) & 0xfffe; a = (int)(short)a; }
        float fa = (float)a * p.scale16;
        CHZ_ROUNDED_F32(fa);
        x = make_float2(fa, 0.f);
        if (i >= p.new_from) { energy += (unsigned)(a * a); clips += (a > 32766 || a < -32766); }
      } else x = p.per == 1 ? make_float2(p.ring[idx], 0.f) : make_float2(p.ring[idx], p.ring[idx + 1]);
      v = cmul(x, p.chirp[i]);
    }
    p.za[i] = v;
  }
  if (p.energy_part != nullptr) {                                    // uniform; every lane takes part in the shuffles
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { energy += __shfl_xor(energy, d); clips += __shfl_xor(clips, d); }
    if ((threadIdx.x & 63) == 0) {
      const int w = (int)blockIdx.x * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
      p.energy_part[w] = energy; p.clip_part[w] = clips;
    }
  }
}
struct BlueMulParams { const float2* zs; const float2* bf; float2* za; SpecLayout lay; long Mz; };
__global__ void __launch_bounds__(256) blue_mul(BlueMulParams p) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < p.Mz; k += stride) {
    const long a = spec_addr(p.lay, k);
    const float2 t = cmul(p.zs[a], p.bf[a]);
    p.za[k] = make_float2(t.x, -t.y);
  }
}
struct BluePostParams { const float2* zs; const float2* chirp; float2* spec; SpecLayout zlay, lay; int bins; float inv; };
__global__ void __launch_bounds__(256) blue_post(BluePostParams p) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < p.bins; k += stride) {
    const float2 y = p.zs[spec_addr(p.zlay, k)];
    p.spec[spec_addr(p.lay, k)] = cmul(make_float2(y.x * p.inv, -y.y * p.inv), p.chirp[k]);
  }
}

// ------------------------------------------------------------------------------
// K5: estimate_noise() of src/radio.c:1783-1866, one workgroup per channel.
// |X|^2 of nbins master bins around |shift| -> LDS, bitonic sort, the 0.10 quantile with linear
// interpolation, mean of the energies <= 1.5 x quantile, times the truncated-exponential correction,
// per Hz.  Keeps the one consumer of the whole spectrum (src/radio.c:1787-1836) on the device.
// ------------------------------------------------------------------------------
// cnrmf(): two products and a sum, each rounded to float
__device__ __forceinline__ float cnrm_unfused(float2 x) {
  float a = x.x * x.x;
  float b = x.y * x.y;
  CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
  return a + b;
}

struct NoiseParams {
  const float2* spec; SpecLayout lay;
  const ChanDesc* desc;   // [nch] this slot's descriptors (the bin shift is read from them)
  double* n0;             // [nch] out: noise density estimate
  int ch0, nch;
  int m_bins, real;       // master bins; real != 0 for a REAL master
  int nbins;              // max(slave bins, Min_noise_bins = 1000)   (:1794-1796)
  int nsort;              // 1024 or 2048: values sorted per channel (64 lanes x 16 or 32 registers)
  unsigned magic; int dpitch;   // bin -> storage index without a division (chan_layout)
  double scale;           // correction / (master bins * front-end sample rate)   (:1840-1844,1863-1865)
  const float* energy;    // EN kernels: |X|^2 of every stored bin (spec_energy), same storage index as spec
  unsigned* hint;         // [nch] or nullptr: exponent + 1 of the channel's last quantile (0: none yet).  A GUESS the kernel verifies with
                          // the counts it needs anyway: a noise floor keeps its binade from block to block, and the search for it -- two
                          // passes for the window's range, four bisection steps -- is half the kernel.  Never part of a result: blocks in
                          // flight on other streams may read an older or a newer guess, and get the same answer either way.
};

// Large banks read every master bin hundreds of times (1.5 M channels x 1000-bin windows over 1.62 M bins): cnrmf() is then taken
// ONCE per bin into an image of floats, and the windows read 4 bytes per bin instead of 8 -- the same float, bit for bit, since it is
// the same three roundings.  The image is in NATURAL bin order (energy[k] for bin k, not the spectrum's pitched storage) with the
// first CHZ_ENERGY_TAIL bins repeated behind the last one: a window is then a plain run of floats from its first bin on, whether
// it wraps around the end of a COMPLEX master or not, and the window kernel has no index arithmetic left per value (round 4: the
// storage index -- a multiply-high, a multiply and three more operations per value -- was a third of noise_est's vector instructions).
// One elementwise pass over the slot (13 MB in, 6.5 MB out), launched by the engine only when the bank is large enough to pay for it.
#define CHZ_ENERGY_TAIL 2048           /* >= the longest window (nsort) */
struct EnergyParams { const float2* spec; float* energy; int bins; int na_off; unsigned magic; int dpitch; };
__global__ void __launch_bounds__(256) spec_energy(EnergyParams p) {
  const int stride = (int)(gridDim.x * blockDim.x);
  const float2* __restrict__ sp = p.spec;
  CHZ_IN_DESC(sdesc, sp);
  float* __restrict__ out = p.energy;
  for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < p.bins; k += stride) {
    const float en = cnrm_unfused(CHZ_LOAD2(sdesc, sp, spec_index(p.na_off, p.magic, p.dpitch, k)));
    out[k] = en;
    if (k < CHZ_ENERGY_TAIL) out[p.bins + k] = en;
  }
}

// Descriptor refresh (round 6): a slot's copy of the gather / fine-tuning / beam descriptors and ISB flags is brought up to the host copy by
// THIS kernel, reading the engine's pinned staging buffer over the host link, in stream order on the slot's lane -- not by hipMemcpyAsync.
// Why: a small host-to-device copy enqueued on a lane right behind the spectrum's device-to-host 2-D copy made the runtime stall the
// CALLING thread for 8-9 ms the first time it happened on a second stream (rocprofv3 --hip-trace of the filter.h drop-in: block 1's
// execute_filter_input took 8.2 ms, every process, profiles/r06_block1_stall.txt); a kernel launch has no such path.  At most 4 segments.
struct PushSeg { void* dst; const void* src; unsigned bytes; unsigned pad; };
struct PushParams { PushSeg seg[4]; int nseg; };
__global__ void __launch_bounds__(256) desc_push(PushParams p) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  for (int g = 0; g < p.nseg; g++) {
    const PushSeg sg = p.seg[g];
    if ((((unsigned long long)sg.dst | (unsigned long long)sg.src | sg.bytes) & 3u) == 0) {
      const unsigned* __restrict__ s = (const unsigned*)sg.src; unsigned* __restrict__ d = (unsigned*)sg.dst;
      for (unsigned i = tid; i < sg.bytes / 4; i += nthr) d[i] = s[i];
    } else {
      const unsigned char* __restrict__ s = (const unsigned char*)sg.src; unsigned char* __restrict__ d = (unsigned char*)sg.dst;
      for (unsigned i = tid; i < sg.bytes; i += nthr) d[i] = s[i];
    }
  }
}

// Reductions over the wavefront for wave-uniform results: DPP inside the rows of 16 lanes, then the four row results through
// v_readlane into scalar registers (no LDS round trips as with ds_bpermute).  On the CPU test emulator: plain shuffles.
#if defined(__HIP_DEVICE_COMPILE__)
#define CHZ_DPP_U32(v, ctrl) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, false))
#endif
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
  { const unsigned o = CHZ_DPP_U32(v, 0xB1); v = o < v ? o : v; }      // quad_perm [1,0,3,2]
  { const unsigned o = CHZ_DPP_U32(v, 0x4E); v = o < v ? o : v; }      // quad_perm [2,3,0,1]
  { const unsigned o = CHZ_DPP_U32(v, 0x141); v = o < v ? o : v; }     // row_half_mirror
  { const unsigned o = CHZ_DPP_U32(v, 0x140); v = o < v ? o : v; }     // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
  return ab < cd ? ab : cd;
#else
  for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(v, d); v = o < v ? o : v; }
  return v;
#endif
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) { return ~wave_min_u32(~v); }
__device__ __forceinline__ int wave_sum_i32(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += (int)CHZ_DPP_U32(v, 0xB1); v += (int)CHZ_DPP_U32(v, 0x4E); v += (int)CHZ_DPP_U32(v, 0x141); v += (int)CHZ_DPP_U32(v, 0x140);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
#else
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
#endif
}
// (the order of the additions differs from a butterfly's: callers that need a fixed order do not use this)
__device__ __forceinline__ double wave_sum_f64(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define CHZ_DPP_F64_STEP(ctrl) { const unsigned long long u = (unsigned long long)__double_as_longlong(v); \
    const unsigned lo = CHZ_DPP_U32((unsigned)u, ctrl), hi = CHZ_DPP_U32((unsigned)(u >> 32), ctrl); \
    v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
  CHZ_DPP_F64_STEP(0xB1) CHZ_DPP_F64_STEP(0x4E) CHZ_DPP_F64_STEP(0x141) CHZ_DPP_F64_STEP(0x140)
#undef CHZ_DPP_F64_STEP
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  double r[4];
  for (int k = 0; k < 4; k++) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * k);
    r[k] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
#else
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
#endif
}
// inclusive prefix sum over the lanes and the wavefront's total
__device__ __forceinline__ int wave_scan_i32(int v, int lane, int& total) {
#if defined(__HIP_DEVICE_COMPILE__)
  // inside each row of 16: row_shr 1, 2, 4, 8 with zero fill
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
  const int row = lane >> 4;
  v += row == 0 ? 0 : (row == 1 ? r0 : (row == 2 ? r0 + r1 : r0 + r1 + r2));
  total = r0 + r1 + r2 + r3;
  return v;
#else
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, (unsigned)d); if (lane >= d) v += o; }
  total = __shfl(v, 63);
  return v;
#endif
}

// One wavefront per channel; the window lives in registers, VPL values per lane.
#define NOISE_KC 4                    /* registers per lane for the values that share the quantile's binade (256 of them) */
template <int VPL, bool EN = false>
__global__ void __launch_bounds__(256) noise_est(NoiseParams p) {
  const int lane = threadIdx.x & 63;
  // the channel index is the same in every lane: telling the compiler keeps the window arithmetic scalar
  const int lc = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (lc >= p.nch) return;                                           // wave-uniform
  const int ch = p.ch0 + lc;
  const int shift = p.desc[ch].shift;
  // first master bin of the window and how many entries the reference fills
  int mbin, n = p.nbins, wrap = 0;
  if (p.real) {
    mbin = (shift < 0 ? -shift : shift) - p.nbins / 2;               // :1812-1816
    if (mbin < 0) mbin = 0; else if (mbin + p.nbins > p.m_bins) mbin = p.m_bins - p.nbins;
  } else {
    mbin = shift - p.nbins / 2;                                      // :1822-1836
    if (mbin < 0) mbin += p.m_bins; else if (mbin >= p.m_bins) mbin -= p.m_bins;
    if (mbin < 0 || mbin >= p.m_bins) { if (lane == 0) p.n0[ch] = 0.0; return; }
    wrap = p.m_bins;
    // the fill stops once the walk arrives at bin m_bins/2 (the +Nyquist seam)
    const int half = p.m_bins / 2;
    const int to_seam = mbin < half ? half - mbin : half + p.m_bins - mbin;
    if (to_seam < n) n = to_seam;
  }
  const float inf = __builtin_huge_valf();
  float v[VPL];
  unsigned mx = 0;                                                   // largest energy of the window (bit pattern), per lane
  // Coalesced loads (which register a bin lands in does not matter to a selection), all issued before
  // any is consumed: entries past n read the window's first bin and are replaced by +inf afterwards.
  // (the storage index of a bin without a division: one multiply-high by the layout's reciprocal, as in chan_ifft)
  if constexpr (EN) {
    // the image is in bin order with the first bins repeated behind the last: the window is a run of floats from mbin on
    // (entries past n read bins beyond the window -- inside the image -- and are replaced)
    const float* __restrict__ en = p.energy;
    CHZ_IN_DESC(edesc, en);
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      v[R] = CHZ_LOAD1(edesc, en, mbin + lane + R * 64);
    });
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      const unsigned u = __float_as_uint(v[R]);
      mx = u > mx ? u : mx;                                           // (an upper bound for the search is all this is: a padding entry may take part)
      if (R * 64 + lane >= n) v[R] = inf;
    });
  } else {
    const float2* __restrict__ sp = p.spec;
    CHZ_IN_DESC(sdesc, sp);
    float2 x[VPL];
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      int i = R * 64 + lane;
      if (i >= n) i = 0;
      int k = mbin + i;
      if (wrap && k >= wrap) k -= wrap;
      x[R] = CHZ_LOAD2(sdesc, sp, spec_index(p.lay.off, p.magic, p.dpitch, k));
    });
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      const float raw = cnrm_unfused(x[R]);
      const unsigned u = __float_as_uint(raw);
      mx = u > mx ? u : mx;
      v[R] = (R * 64 + lane < n) ? raw : inf;
    });
  }
  // quantile(energies, n, 0.10) (:1760-1775): the qi-th and (qi+1)-th smallest energies.  Non-negative floats order like
  // their bit patterns, so the qi-th smallest can be built bit by bit from the top: keep a bit whenever no more than qi
  // values lie strictly below the candidate; counting is one compare per register, a ballot and a scalar popcount.
  // Round 2 ran all 31 steps over all VPL registers: 500 vector compares and 1000 scalar instructions per channel, and the
  // CU's ONE scalar unit was what the kernel waited for (2.4 ns per channel at 1.5 M channels).  Now only the eight exponent
  // bits are found that way.  The values that share the answer's exponent are few -- the 0.10 quantile sits in the thin lower
  // tail -- so they are compacted through LDS into 1..NOISE_KC registers per lane, and the 23 mantissa bits are found
  // over those; a window with more than 64*NOISE_KC values in that one binade carries on over all registers as before.
  const double pos = 0.10 * (double)(n - 1);
  const int qi = (int)floor(pos);
  const double frac = pos - (double)qi;
  unsigned bits[VPL];
  static_for<VPL>([&](auto rr) { constexpr int R = decltype(rr)::value; bits[R] = __float_as_uint(v[R]); });
  // The exponent first, by bisection over the exponents the window actually spans (white noise: some 15 binades, four steps instead
  // of the eight a bit-by-bit search of the field needs): the answer is the largest exponent E with no more than qi values below
  // E << 23, it is not below the smallest value's exponent (nothing lies below that) and not above the largest's.
  int lo = 0, below = 0, mine = 0, total = 0;
  bool guessed = false;
  const unsigned hx = p.hint != nullptr ? (unsigned)__builtin_amdgcn_readfirstlane((int)p.hint[ch]) : 0u;
  if (hx != 0u && hx <= 255u) {
    // last time's binade [t0, t1): right again if no more than qi values lie below t0 and more than qi below t1 -- the counts the
    // compaction needs anyway
    const unsigned t0 = (hx - 1u) << 23, t1 = t0 + (1u << 23);
    int c0 = 0, c1 = 0;
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      const bool a = bits[R] < t0, b = bits[R] < t1;
      c0 += __popcll(__ballot(a)); c1 += __popcll(__ballot(b));
      mine += (b && !a) ? 1 : 0;
    });
    if (c0 <= qi && c1 > qi) { guessed = true; lo = (int)hx - 1; below = c0; total = c1 - c0; }
  }
  if (!guessed) {
    unsigned mn = bits[0];
    static_for<VPL>([&](auto rr) { constexpr int R = decltype(rr)::value; mn = bits[R] < mn ? bits[R] : mn; });
    int hi = (int)(wave_max_u32(mx) >> 23);
    lo = (int)(wave_min_u32(mn) >> 23);
    if (lo > 255) lo = 255;                                            // (NaNs only: they sort above +inf)
    if (hi > 255) hi = 255;
    below = 0;                                                         // values strictly below lo << 23 (wave-uniform)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      const unsigned t = (unsigned)mid << 23;
      int c = 0;
      static_for<VPL>([&](auto rr) { constexpr int R = decltype(rr)::value; c += __popcll(__ballot(bits[R] < t)); });
      if (c <= qi) { lo = mid; below = c; } else hi = mid - 1;         // wave-uniform
    }
  }
  unsigned ans = (unsigned)lo << 23;
  // the binade [ans, ans + 2^23): how many values, and each lane's share of them
  const unsigned top = ans + (1u << 23);                             // (ans has exponent < 255: +inf padding and NaNs sort above every finite value)
  if (!guessed) {
    mine = 0;
    static_for<VPL>([&](auto rr) { constexpr int R = decltype(rr)::value; mine += (bits[R] >= ans && bits[R] < top) ? 1 : 0; });
  }
  int scan_total;
  const int incl = wave_scan_i32(mine, lane, scan_total);            // inclusive scan over the lanes
  if (!guessed) total = scan_total;
  if (p.hint != nullptr && !guessed && lane == 0) p.hint[ch] = (unsigned)lo + 1u;
  const int r = qi - below;                                          // rank of the answer inside the binade, 0 <= r < total
  unsigned cur = ans;
  int c_le;                                                          // values <= the answer, all of the window
  unsigned above = 0x7f800000u;                                      // the smallest value above the answer
  if (total <= 64 * NOISE_KC) {
    __shared__ unsigned cand_all[4][64 * NOISE_KC];
    unsigned* cand = cand_all[(threadIdx.x >> 6) & 3];
    int at = incl - mine;
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      if (bits[R] >= ans && bits[R] < top) { cand[at] = bits[R]; at++; }
    });
    CHZ_WAVE_SYNC();
    unsigned cv[NOISE_KC];
    static_for<NOISE_KC>([&](auto jj) {
      constexpr int J = decltype(jj)::value;
      cv[J] = (J * 64 + lane < total) ? cand[J * 64 + lane] : 0xffffffffu;
    });
    int cl = 0;
    if (total <= 64) {                                               // the usual case: one value per lane
      for (int bit = 22; bit >= 0; --bit) {
        const unsigned t = cur | (1u << bit);
        if (__popcll(__ballot(cv[0] < t)) <= r) cur = t;
      }
      cl = __popcll(__ballot(cv[0] <= cur));
      if (cv[0] > cur && cv[0] < above) above = cv[0];
    } else if (total <= 128) {                                       // white noise puts 50 to 90 values into the quantile's binade
      for (int bit = 22; bit >= 0; --bit) {
        const unsigned t = cur | (1u << bit);
        if (__popcll(__ballot(cv[0] < t)) + __popcll(__ballot(cv[1] < t)) <= r) cur = t;
      }
      static_for<2>([&](auto jj) {
        constexpr int J = decltype(jj)::value;
        cl += __popcll(__ballot(cv[J] <= cur)); if (cv[J] > cur && cv[J] < above) above = cv[J];
      });
    } else {
      for (int bit = 22; bit >= 0; --bit) {
        const unsigned t = cur | (1u << bit);
        int c = 0;
        static_for<NOISE_KC>([&](auto jj) { constexpr int J = decltype(jj)::value; c += __popcll(__ballot(cv[J] < t)); });   // (unused registers hold 0xffffffff)
        if (c <= r) cur = t;
      }
      static_for<NOISE_KC>([&](auto jj) {
        constexpr int J = decltype(jj)::value;
        cl += __popcll(__ballot(cv[J] <= cur)); if (cv[J] > cur && cv[J] < above) above = cv[J];
      });
    }
    c_le = below + cl;
    if (frac != 0.0 && c_le < qi + 2 && cl == total) {
      // the next order statistic lies beyond this binade (rare): the smallest value of the rest of the window
      static_for<VPL>([&](auto rr) { constexpr int R = decltype(rr)::value; if (bits[R] >= top && bits[R] < above) above = bits[R]; });
    }
  } else {
    for (int bit = 22; bit >= 0; --bit) {
      const unsigned t = cur | (1u << bit);
      int c = 0;
      static_for<VPL>([&](auto rr) { constexpr int R = decltype(rr)::value; c += __popcll(__ballot(bits[R] < t)); });
      if (c <= qi) cur = t;
    }
    c_le = 0;
    static_for<VPL>([&](auto rr) {
      constexpr int R = decltype(rr)::value;
      c_le += __popcll(__ballot(bits[R] <= cur));
      if (bits[R] > cur && bits[R] < above) above = bits[R];
    });
  }
  ans = cur;
  const double q1 = (double)__uint_as_float(ans);
  double q = q1;
  if (frac != 0.0) {
    // the next order statistic: q1 again if it occurs more than once beyond rank qi, else the smallest value above it
    above = wave_min_u32(above);
    const double q2 = c_le >= qi + 2 ? q1 : (double)__uint_as_float(above);
    double fq = frac * (q2 - q1);                                    // :1773, product rounded before the sum
    CHZ_ROUNDED_F64(fq);
    q = q1 + fq;
  }
  const double cut = 1.5 * q;
  // (double)x <= cut for a float x  <=>  x <= the largest float not above cut: one integer compare per value instead of a
  // conversion and a double compare
  float cf = (float)cut;
  if ((double)cf > cut) cf = __uint_as_float(__float_as_uint(cf) - 1u);        // cut > 0 here, or cut == 0 == cf
  const unsigned cb = cut >= 0.0 ? __float_as_uint(cf) : 0u;
  const bool none = !(cut >= 0.0);
  double e = 0.0; int cnt = 0;
  static_for<VPL>([&](auto rr) {
    constexpr int R = decltype(rr)::value;
    if (!none && bits[R] <= cb) { e += (double)v[R]; cnt++; }        // the +inf padding never qualifies
  });
  e = wave_sum_f64(e); cnt = wave_sum_i32(cnt);
  if (lane == 0) p.n0[ch] = cnt ? e / (double)cnt * p.scale : 0.0;
}

// ------------------------------------------------------------------------------
// the linear demodulator's per-block work behind the fine-tuned channel outputs
// (demod_linear, src/linear.c:56-375): the PLL of the coherent modes (:76-153), noise smoothing (src/radio.c:1466-1473), the
// post-detection shift oscillator (:168-172), block AGC (:177-234), the final demodulation pass with the per-sample
// gain ramp (:236-311), the SNR squelch sequencer (:313-366), and PCM packing (src/import.h:88-118 via send_output,
// src/audio.c:117-133).  What leaves the device per channel and block is the packed PCM (480 B for a 12 kHz mono
// S16 channel instead of 1920 B of complex baseband) and a small status record.
// One lane per channel, every loop in the reference's own order: gain *= gain_change and the carrier-removal filter
// are recurrences over the samples, and doing them as the reference does keeps the result bit-for-bit that of the
// restatement (oracle/chz_oracle.c:chzo_lindemod_block), which is pinned to the reference's linear.c itself.
// The state is a recurrence over BLOCKS: the engine runs these kernels on one in-order stream.
// ------------------------------------------------------------------------------
enum { CHZ_PCM_S16BE_K = 0, CHZ_PCM_S16LE_K = 1, CHZ_PCM_F32LE_K = 2, CHZ_PCM_F32BE_K = 3, CHZ_PCM_MULAW_K = 4, CHZ_PCM_ALAW_K = 5, CHZ_PCM_F16LE_K = 6, CHZ_PCM_F16BE_K = 7 };
struct DemodChan {               // per channel, set by the host (names: the chan_t members src/linear.c reads)
  int channels, env, agc, encoding, snr_squelch, squelch_tail, tuned, on;
  double samprate, headroom, threshold, recovery_rate, hangtime, dc_alpha, bandwidth, squelch_open, squelch_close;
  double osc_phase0, osc_freq;   // chan->shift as a closed form in the block number: phase (cycles) at sample 0 of block osc_job0
  unsigned osc_job0; int kind;   // kind 0: linear demodulator (src/linear.c), 1: FM (src/fm.c)
  double deemph_rate, deemph_gain, threshold_extend;      // FM: chan->fm.rate, chan->fm.gain, chan->fm.threshold
  int pll_enable, pll_square;    // chan->pll.enable / .square (linear: src/linear.c:83-153; FM: src/fm.c:176-203)
  double pll_loop_bw;            // chan->pll.loop_bw, Hz
  double tone_freq;              // FM: chan->fm.tone_freq (0 = no PL tone squelch, src/fm.c:264-311)
  double g_coeff, g_cfr, g_cfi;  // init_goertzel(tone_freq / samprate) (src/iir.c:32-39), computed by the host
  double recov_ps;               // pow(recovery_rate, 1 / samprate) (src/linear.c:231), computed by the host: a constant of the channel
};
// State of the coherent modes and of the PL-tone squelch; only channels that use them touch it (one lane, sequentially).
struct PllState { unsigned vco_phase; int vco_step, wraps, lock, lock_count, pad;         // struct pll (src/osc.h:21-32) + chan->pll.lock / .lock_count
                  double bw, damping, lower, upper, u, phi, K1, K2; };
struct DemodExt { PllState pll; double pll_snr, pll_cphase, foffset;                       // chan->pll.snr / .cphase, chan->sig.foffset (linear)
                  double g_s0, g_s1, old_pl_phase, tone_deviation;                         // Goertzel state, src/fm.c:60, chan->fm.tone_deviation
                  int pll_rotations, pl_sample_count, tone_mute, pad;
                  // hand-over between the passes of an FM channel whose PLL demodulator / PL-tone detector runs one channel per lane
                  double fm_snr, fm_noise;                                                 // fm_front_k -> fm_pll_lanes, demod_linear_tail, fm_finish
                  int fm_go, fm_stage; };                                                  // squelch open this block; 1 = waiting for the tone decision
struct DemodState { double gain, am_dc, n0; int hangcount, squelch_state, squelch_open, pll_was_on;   // pll_was_on: chan->pll.was_on (FM, src/fm.c:178-184,209)
                    double pm_re, pm_im, deemph_state, foffset, pdeviation; };   // FM: phase_memory, de-emphasis state, chan->sig.foffset, chan->fm.pdeviation
struct DemodStatus { int frame, mute, squelch_state, pll_lock; double output_power, gain, n0, snr, foffset, pdeviation;   // frame 0 = PCM present, 1 = silence
                     double pll_snr, pll_cphase, tone_deviation; int pll_rotations, tone_mute; };
struct DemodParams {
  const float2* in;          // [cap][olen] this slot's channel outputs (after fine tuning)
  const double* power;       // [cap] this slot's bb_power
  const double* n0;          // [cap] this slot's noise estimates
  const DemodChan* chan;     // [cap]
  DemodState* state;         // [cap]
  DemodExt* ext;             // [cap] PLL / tone-squelch state (read only by channels that enable them)
  DemodStatus* status;       // [cap] this slot
  unsigned char* flags;      // [cap] this slot: the status in one byte (CHZ_FLAG_*), what send_output()'s caller needs every block
  unsigned char* pcm;        // [cap][pcm_stride] this slot
  int ch0, nch, olen, pcm_stride;
  unsigned job;
  double blocktime, power_alpha;
  double fm_alpha;           // -expm1(-blocktime / 1 s): the smoothing constant of FM's frequency-offset estimate (src/fm.c:55); filled in by launch_demod
  int lin_lanes;             // the linear demodulator's channels are served by demod_lin_lanes (one channel per lane); set by launch_demod
  int fm_lanes;              // the FM channels without the PLL demodulator are served by demod_fm_lanes (needs `mix`)
  int wave_any;              // the bank has channels demod_linear_tail must serve (FM; PLL channels without the scratch block)
  int lin_pll, fm_pll, fm_tone;   // the bank has channels with a carrier PLL (linear) / the PLL demodulator (FM) / a PL-tone squelch (FM): which
                             // of the lane-per-channel passes launch_demod adds; they need `mix`
  double* agc_peak;          // [cap] this slot, or nullptr: the block's largest 2 ms slice energy (the AGC's first look at the block,
                             // src/linear.c:177-203), left by whoever had the block's samples in hand last: chan_ifft (peak_chan: channels outside the
                             // coherent modes) or pll_lanes (peak_pll: the mixed block of a coherent-mode channel); not valid for a channel with a
                             // post-detection shift, which rotates the block once more
  int peak_chan, peak_pll;
  float2* mix;               // [cap][olen] or nullptr: the coherent modes' blocks after their PLL (written by pll_lanes, one CHANNEL PER LANE);
                             // nullptr: lane 0 of each channel's wavefront walks the block inside demod_linear_tail (round 2's way)
};

// G.711 companding as send_output() applies it (float_to_mulaw / float_to_alaw, src/rtp.c:459-483,500-533): clamp, 16-bit
// sign/magnitude, clip at 32635, segment = position of the leading one, 4 mantissa bits below it
__device__ __forceinline__ unsigned char demod_g711(float v, bool alaw) {
  v = v > 1.0f ? 1.0f : v < -1.0f ? -1.0f : v;
  const int sample = (int)rintf(ldexpf(v, 15));
  const int sign = sample < 0;
  int pcm = sign ? -sample : sample;
  if (pcm > 32635) pcm = 32635;
  if (!alaw) pcm += 0x84;
  int exponent = (alaw && pcm < 256) ? 0 : (31 - __builtin_clz((unsigned)pcm)) - 7;
  exponent = exponent < 0 ? 0 : exponent > 7 ? 7 : exponent;
  const int mantissa = (alaw && exponent == 0) ? (pcm >> 4) & 0x0F : (pcm >> (exponent + 3)) & 0x0F;
  const unsigned char code = (unsigned char)((exponent << 4) | mantissa);
  return alaw ? (unsigned char)(code ^ (sign ? 0xD5 : 0x55)) : (unsigned char)~(code | (sign << 7));
}
// one byte per channel and block: bit 0 frame has no samples (send_output(chan, NULL, ..)), bit 1 mute, bit 2 PLL locked, bit 3 tone squelch muting
__device__ __forceinline__ void demod_publish(const DemodParams& p, int ch, const DemodStatus& r) {
  p.status[ch] = r;
  if (p.flags != nullptr) p.flags[ch] = (unsigned char)((r.frame & 1) | ((r.mute & 1) << 1) | ((r.pll_lock & 1) << 2) | ((r.tone_mute & 1) << 3));
}
// float -> IEEE binary16 bits, round to nearest even: what `float16_t temp_float = in[i]` of export_f16_* does (src/import.h:140-157).
// The device has the conversion in hardware (v_cvt_f16_f32, RNE); the CPU test emulator (g++ 11: no _Float16) does it by hand.
__device__ __forceinline__ unsigned short demod_f16_bits(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const _Float16 h = (_Float16)v;
  unsigned short u; __builtin_memcpy(&u, &h, 2);
  return u;
#else
  const unsigned x = __float_as_uint(v), sign = (x >> 16) & 0x8000u, mag = x & 0x7fffffffu;
  if (mag >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (mag > 0x7f800000u ? 0x200u | ((mag >> 13) & 0x3ffu) : 0u));   // inf / NaN
  if (mag >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);                        // rounds to or beyond 65520: inf
  if (mag < 0x33000001u) return (unsigned short)sign;                                      // at or below half the smallest subnormal: +-0
  const int e = (int)(mag >> 23) - 127;
  unsigned m = (mag & 0x7fffffu) | 0x800000u;                                              // 24-bit significand
  const int shift = e < -14 ? 13 + (-14 - e) : 13;                                         // subnormal halves lose more bits
  const unsigned keep = m >> shift, rest = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  unsigned r = keep + ((rest > half || (rest == half && (keep & 1u))) ? 1u : 0u);
  const unsigned hb = e < -14 ? r : ((unsigned)(e + 15) << 10) + (r - 0x400u);              // a carry out of the significand lands in the exponent
  return (unsigned short)(sign | hb);
#endif
}
// export_s16_*() for one sample (src/import.h:90-94): what demod_put stores for the S16 encodings
__device__ __forceinline__ unsigned short demod_s16(float v, bool big_endian) {
  float t = ldexpf(v, 15);
  t = t > 32767.0f ? 32767.0f : t < -32767.0f ? -32767.0f : t;
  const int q = (int)rintf(t);                                         // lrintf: to nearest, ties to even
  unsigned short u = (unsigned short)(short)q;
  if (big_endian) u = (unsigned short)((u >> 8) | (u << 8));
  return u;
}
__device__ __forceinline__ void demod_put(unsigned char* o, int enc, int idx, float v) {
  if (enc == CHZ_PCM_F16LE_K || enc == CHZ_PCM_F16BE_K) {
    unsigned short u = demod_f16_bits(v);
    if (enc == CHZ_PCM_F16BE_K) u = (unsigned short)((u >> 8) | (u << 8));
    reinterpret_cast<unsigned short*>(o)[idx] = u;
  } else if (enc == CHZ_PCM_MULAW_K || enc == CHZ_PCM_ALAW_K) {
    o[idx] = demod_g711(v, enc == CHZ_PCM_ALAW_K);
  } else if (enc == CHZ_PCM_S16BE_K || enc == CHZ_PCM_S16LE_K) {
    float t = ldexpf(v, 15);                                           // src/import.h:90-94
    t = t > 32767.0f ? 32767.0f : t < -32767.0f ? -32767.0f : t;
    const int q = (int)rintf(t);                                       // lrintf: to nearest, ties to even
    unsigned short u = (unsigned short)(short)q;
    if (enc == CHZ_PCM_S16BE_K) u = (unsigned short)((u >> 8) | (u << 8));
    reinterpret_cast<unsigned short*>(o)[idx] = u;
  } else {
    unsigned u = __float_as_uint(v);
    if (enc == CHZ_PCM_F32BE_K) u = __builtin_bswap32(u);
    reinterpret_cast<unsigned*>(o)[idx] = u;
  }
}
__device__ __forceinline__ float demod_cabsf(float2 x) {
  float a = x.x * x.x, b = x.y * x.y;
  CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
  return sqrtf(a + b);
}

// One WAVEFRONT per channel.  Lane l owns the SEG = ceil(N/64) consecutive samples [l*SEG, (l+1)*SEG): its loads are
// one contiguous run, a wavefront reads the channel's whole row as full cache lines and writes its PCM the same way.
//   * 2 ms AGC slices: per-sample energies go through LDS and lane s adds up slice s in the reference's own order;
//   * the gain ramp gain *= gain_change (src/linear.c:253,271,...) starts in lane l at gain * gain_change^(l*SEG) (one pow per
//     lane) and is then multiplied along as the reference does; the block's final gain is the last lane's;
//   * the carrier-removal filter am_dc += alpha (s - am_dc) (src/linear.c:257-260) is a first-order linear recurrence: each
//     lane reduces its samples to an affine map, a wavefront scan composes the maps, and the lane replays its samples from
//     the right starting value;
//   * output power: per-lane sums, then a butterfly over the wavefront.
// Against the restatement's strictly sequential loops these reorderings differ by a few ulp of a DOUBLE (1e-15 relative),
// far below the float / int16 the samples are rounded to; everything decision-making (AGC branches, squelch sequencer,
// packing) is the reference's statement for statement.
// wave-wide sum / max / min of doubles, the same value in every lane: DPP inside the rows of 16, v_readlane across them (no LDS round
// trips); on the CPU test emulator plain shuffles
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_f64(v); }
template <bool MAX> __device__ __forceinline__ double wave_ext_f64(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
#define CHZ_DPP_F64_EXT(ctrl) { const unsigned long long u = (unsigned long long)__double_as_longlong(v); \
    const unsigned lo = CHZ_DPP_U32((unsigned)u, ctrl), hi = CHZ_DPP_U32((unsigned)(u >> 32), ctrl); \
    const double o = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); v = MAX ? (o > v ? o : v) : (o < v ? o : v); }
  CHZ_DPP_F64_EXT(0xB1) CHZ_DPP_F64_EXT(0x4E) CHZ_DPP_F64_EXT(0x141) CHZ_DPP_F64_EXT(0x140)
#undef CHZ_DPP_F64_EXT
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  double r = v;
  for (int k = 0; k < 4; k++) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 16 * k), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 16 * k);
    const double o = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    r = k == 0 ? o : (MAX ? (o > r ? o : r) : (o < r ? o : r));
  }
  return r;
#else
  for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(v, d); v = MAX ? (o > v ? o : v) : (o < v ? o : v); }
  return v;
#endif
}
__device__ __forceinline__ double wave_max(double v) { return wave_ext_f64<true>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_ext_f64<false>(v); }
// fm_snr() with its Bessel series (src/misc.c:414-468): amplitude mean^2/variance of a Rice process -> signal-to-noise ratio
__device__ inline double fm_i0(double z) { double t = 0.25 * z * z, sum = 1 + t, term = t;
  for (int k = 2; k < 40; k++) { term *= t / (double)(k * k); sum += term; if (term < 1e-12 * sum) break; } return sum; }
__device__ inline double fm_i1(double z) { double t = 0.25 * z * z, term = 0.5 * t, sum = 1 + term;
  for (int k = 2; k < 40; k++) { term *= t / (double)(k * (k + 1)); sum += term; if (term < 1e-12 * sum) break; } return 0.5 * z * sum; }
__device__ inline double fm_xi(double thetasq) {
  double t = (2 + thetasq) * fm_i0(0.25 * thetasq) + thetasq * fm_i1(0.25 * thetasq);
  t *= t;
  return 2 + thetasq - (0.125 * M_PI) * exp(-0.5 * thetasq) * t;
}
__device__ inline double fm_snr_dev(double r) {
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  double thetasq = r;
  for (int i = 0; i < 10; i++) {
    const double o = thetasq;
    thetasq = fm_xi(thetasq) * (1 + r) - 2;
    if (fabs(thetasq - o) <= 0.01) break;
  }
  return thetasq;
}

// ---- the PLL of the coherent modes (src/osc.c:75-205).  A PLL is a recurrence through a non-linear phase detector: it runs
// sample by sample on ONE lane of the channel's wavefront (the block sits in LDS); channels are independent, so a launch still
// keeps one lane busy per PLL channel.  nco(): the reference indexes a table of sin(pi/2 * i/1024); the same values are
// computed here (sincospi, within an ulp of a double of the table's), the second-order interpolation step is the reference's.
__device__ inline void pll_nco(unsigned accum, double& s, double& c) {
  const unsigned fract = accum & ((1u << 20) - 1);
  unsigned tab = (accum >> 20) & 1023u;
  unsigned quad = accum >> 30;
  tab = (quad & 1) ? 1024 - tab : tab;
  double ts, tc;
  sincospi((double)tab * (1.0 / 2048.0), &ts, &tc);                        // Lookup[tab], Lookup[1024 - tab]
  const double sine = (quad & 2) ? -ts : ts;
  quad++;
  const double cosine = (quad & 2) ? -tc : tc;
  const double diff = 2 * M_PI * ldexp((double)fract, -32);
  const double cdiff = cosine * diff, sdiff = sine * diff;
  s = sine + cdiff - 0.5 * sdiff * diff;
  c = cosine - sdiff - 0.5 * cdiff * diff;
}
__host__ __device__ inline void pll_set_params(PllState& q, double bw, double damping) {     // src/osc.c:152-167
  if (bw == 0 || (bw == q.bw && damping == q.damping)) return;
  const double denom = damping + 1.0 / (4.0 * damping);
  const double theta = 4.0 * M_PI * fabs(bw) / denom;
  q.bw = bw; q.damping = damping;
  const double D = 1.0 + 2.0 * damping * theta + theta * theta;
  q.K1 = 4.0 * damping * theta / D;
  q.K2 = 4.0 * theta * theta / D;
}
__host__ __device__ inline void pll_init(PllState& q) {                                       // src/osc.c:130-136
  q.vco_phase = 0; q.vco_step = 0; q.wraps = 0; q.bw = 0; q.damping = 0; q.u = 0; q.phi = 0; q.K1 = 0; q.K2 = 0;
  q.lower = -0.5; q.upper = +0.5;
  pll_set_params(q, 0.01, M_SQRT1_2);
}
__device__ inline double pll_run(PllState& q, double phase) {                                 // src/osc.c:174-205
  double u_new = q.u + q.K2 * phase;
  double dphi = u_new + q.K1 * phase;
  if (dphi > q.upper) { dphi = q.upper; if (phase > 0) u_new = q.u; }
  else if (dphi < q.lower) { dphi = q.lower; if (phase < 0) u_new = q.u; }
  q.u = u_new;
  q.phi += dphi;
  if (q.phi > 1) { q.phi -= 1; q.wraps++; }
  else if (q.phi < -1) { q.phi += 1; q.wraps--; }
  q.vco_step = (int)ldexp(dphi, 32);
  q.vco_phase += (unsigned)q.vco_step;
  return q.u;
}

// demod_fm()'s per-block work (src/fm.c:19-345), one wavefront per channel, lane l owning SEG consecutive samples.  esh[] is
// per-lane scratch (every lane reads only what it wrote) except around the two sequential stages -- the PLL demodulator
// (:176-203) and the PL-tone detector (:264-311) -- which lane 0 runs over the whole block between wavefront syncs.
// ---- demod_fm() in pieces, so that the same statements serve the one-kernel path and the split one (PLL demodulator and PL-tone
// detector at one channel per lane, below): every piece is the reference's own order of operations.
struct FmFront { double fmsnr, noise; };
// :55-155 noise smoothing, both SNR estimators, the squelch sequencer.  Advances st.n0 and st.squelch_state.
__device__ __forceinline__ FmFront fm_front(const DemodParams& p, const DemodChan& c, DemodState& st, int ch, const float2* __restrict__ x,
                                            int n0, int cnt, int N, double* esh) {
  const double bb_power = p.power[ch];
  const double est = p.n0[ch];
  if (st.n0 != st.n0) st.n0 = est;
  else { const double diff = est - st.n0; st.n0 += p.power_alpha * diff; }
  FmFront f;
  f.noise = st.n0 * c.bandwidth;                                                // :101
  const double snr = f.noise == 0 ? __builtin_huge_val() : (bb_power / f.noise) - 1.0;
  if (c.snr_squelch || (st.squelch_state <= 0 && snr < c.squelch_close)) {
    f.fmsnr = snr;
  } else {                                                                      // :110-129 amplitude variance
    double part = 0.0;
    for (int i = 0; i < cnt; i++) { const double a = (double)demod_cabsf(x[n0 + i]); esh[n0 + i] = a; part += a; }
    const double avg = wave_sum(part) / N;
    part = 0.0;
    for (int i = 0; i < cnt; i++) { const double dlt = esh[n0 + i] - avg; part += dlt * dlt; }
    const double var = wave_sum(part);
    const double s2 = fm_snr_dev(avg * avg * (N - 1) / var);
    f.fmsnr = s2 > 0.0 ? s2 : 0.0;
  }
  const int smax = c.squelch_tail + 5;                                          // :149-155
  if (f.fmsnr >= c.squelch_open) st.squelch_state = smax;
  else if (st.squelch_state > 0 && (f.fmsnr < c.squelch_close || st.squelch_state < smax)) st.squelch_state--;
  return f;
}
// one sample of the PLL demodulator (:185-201)
__device__ __forceinline__ float fm_pll_sample(PllState& q, float2 v, double pdev, bool extend, double beta, double noise) {
  double sn, cs; pll_nco(q.vco_phase, sn, cs);
  const double br = v.x, bi = v.y;
  const double sr = br * cs + bi * sn, si = bi * cs - br * sn;                  // buffer[n] * conj(vco)
  double phase = M_1_PI * atan2(si, sr);
  if (extend) {
    if (fabs(phase) > pdev) phase = copysign(pdev, phase);
    float a = v.x * v.x, b = v.y * v.y; CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
    double pw = (double)(a + b);
    if (pw > 0) { pw /= (pw + beta * noise); phase *= pw; }
    else phase = 0;
  }
  return (float)(2 * pll_run(q, phase));
}
// the PL-tone detector's state and one sample of it (:264-304)
struct FmTone { double s0, s1, old_phase, tdev; int count, tmute; };
__device__ __forceinline__ void fm_tone_sample(FmTone& g, const DemodChan& c, double xin, int isamprate, int integrate) {
  { const double t = xin + c.g_coeff * g.s0 - g.s1; g.s1 = g.s0; g.s0 = t; }   // update_goertzel
  if (++g.count >= integrate) {
    { const double t = 0.0 + c.g_coeff * g.s0 - g.s1; g.s1 = g.s0; g.s0 = t; } // output_goertzel: one zero sample
    const double cre = g.s0 - c.g_cfr * g.s1, cim = -c.g_cfi * g.s1;
    const double gm = sqrt(cre * cre + cim * cim) / g.count;
    g.tdev = isamprate * gm;
    const double ph = atan2(cim, cre) / (2 * M_PI);
    g.old_phase += c.tone_freq * g.count / isamprate;
    double ip;
    double np = 2 * modf(ph - g.old_phase, &ip);
    g.old_phase = ph;
    np = np < -1 ? np + 2 : np > 1 ? np - 2 : np;
    g.tmute = g.tdev < 250 || fabs(np) > .10;
    g.s0 = 0.0; g.s1 = 0.0; g.count = 0;
  }
}
// :312-334 de-emphasis (a scan of affine maps over the lanes' segments), gain, PCM, the status record.  esh[] holds the baseband
// after DC removal.
__device__ __forceinline__ void fm_deemph_output(const DemodParams& p, const DemodChan& c, DemodState& st, DemodStatus& r, int ch, int lane,
                                                 double* esh, int n0, int cnt, int N) {
  unsigned char* __restrict__ o = p.pcm + (size_t)ch * p.pcm_stride;
  const bool pm = c.deemph_rate != 0;
  double y_in = st.deemph_state;
  if (pm) {
    double A = 1.0, B = 0.0;
    for (int i = 0; i < cnt; i++) {
      const float b = (float)esh[n0 + i];
      A *= (1.0 - c.deemph_rate); B = (1.0 - c.deemph_rate) * B + c.deemph_rate * (c.deemph_gain * (double)b);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double Ap = __shfl_up(A, d), Bp = __shfl_up(B, d);
      if (lane >= d) { B = A * Bp + B; A = A * Ap; }
    }
    const double Ae = __shfl_up(A, 1), Be = __shfl_up(B, 1);
    y_in = lane == 0 ? st.deemph_state : Ae * st.deemph_state + Be;
    st.deemph_state = __shfl(A, 63) * st.deemph_state + __shfl(B, 63);
  }
  const double gain = (2 * c.headroom * c.samprate) / c.bandwidth;              // :325
  double part = 0.0;
  {
    double y = y_in;
    for (int i = 0; i < cnt; i++) {
      float b = (float)esh[n0 + i];
      if (pm) { y += c.deemph_rate * (c.deemph_gain * (double)b - y); b = (float)y; }
      const double sgn = gain * (double)b;
      part += sgn * sgn;
      demod_put(o, c.encoding, n0 + i, (float)sgn);
    }
  }
  r.frame = 0; r.mute = 0; r.gain = gain; r.output_power = wave_sum(part) / N; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
  if (lane == 0) { demod_publish(p, ch, r); p.state[ch] = st; }
}

__device__ __forceinline__ void demod_fm_wave(const DemodParams& p, const DemodChan& c, DemodState st, int ch, int lane, double* esh, float2* xs) {
  const int N = p.olen;
  const int SEG = (N + 63) >> 6;
  const int n0 = lane * SEG;
  const int cnt = n0 >= N ? 0 : (N - n0 < SEG ? N - n0 : SEG);
  const float2* __restrict__ x = p.in + (size_t)ch * N;
  const double samprate = c.samprate, devmax = 5000.0, beta = 0.5;             // src/fm.c:43,103
  const bool tone = c.tone_freq != 0, pll = c.pll_enable != 0;                  // wave-uniform
  const bool pll_split = pll && p.mix != nullptr && p.fm_pll != 0;              // fm_front_k + fm_pll_lanes have run for this channel
  const bool tone_split = tone && p.mix != nullptr && p.fm_tone != 0;           // fm_tone_lanes + fm_finish will
  DemodExt* __restrict__ ext = p.ext + ch;                                      // touched only with pll / tone
  float* __restrict__ mixf = reinterpret_cast<float*>(p.mix) + (size_t)ch * 2 * N;   // [N] PLL baseband, [N] the tone detector's input
  const double alpha = p.fm_alpha;                                              // -expm1(-blocktime / 1.0), :55 (launch_demod)
  FmFront f;
  if (pll_split) { f.fmsnr = ext->fm_snr; f.noise = ext->fm_noise; }           // (st was advanced by fm_front_k)
  else f = fm_front(p, c, st, ch, x, n0, cnt, N, esh);
  const double fmsnr = f.fmsnr, noise = f.noise;
  const int smax = c.squelch_tail + 5;
  DemodStatus r;
  r.pll_lock = 0; r.pll_snr = 0.0; r.pll_cphase = 0.0; r.pll_rotations = 0; r.tone_deviation = 0.0; r.tone_mute = 0;
  r.n0 = st.n0; r.snr = fmsnr; r.squelch_state = st.squelch_state; r.gain = 0.0;
  if (st.squelch_state <= 4) {                                                  // :157-173
    if (st.squelch_state >= 1) { st.pm_re = 0.0; st.pm_im = 0.0; }
    r.frame = 1; r.mute = st.squelch_state == 0; r.output_power = 0.0; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
    if (lane == 0) {
      if (tone) {
        if (st.squelch_state == 4) { ext->g_s0 = 0.0; ext->g_s1 = 0.0; }        // reset_goertzel
        if (st.squelch_state >= 1) ext->pl_sample_count = 0;
        r.tone_deviation = ext->tone_deviation; r.tone_mute = ext->tone_mute;
      }
      demod_publish(p, ch, r); p.state[ch] = st;
    }
    return;
  }
  if (pll_split) {
    for (int i = 0; i < cnt; i++) esh[n0 + i] = (double)mixf[n0 + i];          // fm_pll_lanes has walked the block
    st.pll_was_on = 1;
  } else if (pll) {
    // :176-203 PLL demodulator: the block goes to LDS, lane 0 walks it
    for (int i = 0; i < cnt; i++) xs[n0 + i] = x[n0 + i];
    CHZ_WAVE_SYNC();
    if (lane == 0) {
      PllState q = ext->pll;
      const int isamprate = (int)samprate;
      const double pdev = devmax / isamprate;
      if (!st.pll_was_on) {
        pll_init(q);
        pll_set_params(q, 500.0 / isamprate, M_SQRT1_2);
        q.lower = -pdev; q.upper = +pdev;
      }
      for (int n = 0; n < N; n++) esh[n] = (double)fm_pll_sample(q, xs[n], pdev, c.threshold_extend != 0, beta, noise);
      ext->pll = q;
    }
    st.pll_was_on = 1;
    CHZ_WAVE_SYNC();
  } else {
    st.pll_was_on = 0;                                                          // :209
    // :204-231 discriminator: phase of x[n] * conj(x[n-1]); the sample before the block is phase_memory
    for (int i = 0; i < cnt; i++) {
      const int n = n0 + i;
      const float2 v = x[n];
      double pr, pi, p0;
      if (n == 0) { pr = st.pm_re; pi = st.pm_im; p0 = pr * pr + pi * pi; }
      else { const float2 w = x[n - 1]; pr = w.x; pi = w.y; float a = w.x * w.x, b = w.y * w.y; CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b); p0 = (double)(a + b); }
      const double br = v.x, bi = v.y;
      const double sr = br * pr + bi * pi, si = bi * pr - br * pi;
      double phase = M_1_PI * atan2(si, sr);
      if (c.threshold_extend != 0) {
        if (fabs(phase) > devmax / samprate) phase = copysign(devmax / samprate, phase);
        if (p0 > 0) p0 /= (p0 + beta * noise);
        float a = v.x * v.x, b = v.y * v.y; CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
        double p1 = (double)(a + b);
        if (p1 > 0) p1 /= (p1 + beta * noise);
        phase *= p0 * p1;
      }
      esh[n] = (double)(float)phase;
    }
  }
  double psum = 0.0, pmax = 0.0, pmin = 0.0;
  for (int i = 0; i < cnt; i++) {
    const double bbv = esh[n0 + i];
    psum += bbv;
    if (bbv > pmax) pmax = bbv;
    if (bbv < pmin) pmin = bbv;
  }
  {
    const float2 lastv = x[N - 1];                                              // phase_memory = the block's last sample
    st.pm_re = lastv.x; st.pm_im = lastv.y;
  }
  if (st.squelch_state == smax) {                                               // :232-256
    double foff = wave_sum(psum) * (samprate * 0.5 / N);
    double ppos = wave_max(pmax), pneg = wave_min(pmin);
    st.foffset += alpha * (foff - st.foffset);
    ppos *= samprate * 0.5; pneg *= samprate * 0.5;
    ppos -= st.foffset; pneg -= st.foffset;
    st.pdeviation = ppos > -pneg ? ppos : -pneg;
  }
  if (c.deemph_rate != 0) {                                                     // :258-263 DC removal (with the de-emphasis, :312-320)
    const float dc = (float)(2 * st.foffset / samprate);
    for (int i = 0; i < cnt; i++) { const float b = (float)esh[n0 + i] - dc; esh[n0 + i] = (double)b; }
  }
  if (tone_split) {
    // the tone detector runs one channel per lane in fm_tone_lanes; fm_finish takes the block from there
    for (int i = 0; i < cnt; i++) mixf[N + n0 + i] = (float)esh[n0 + i];
    if (lane == 0) { ext->fm_snr = fmsnr; ext->fm_stage = 1; p.state[ch] = st; }
    return;
  }
  if (tone) {
    // :264-311 PL / CTCSS tone squelch on the baseband after DC removal and before de-emphasis: a Goertzel detector integrated over
    // 0.24 s (12 blocks), the decision taken where the count runs out -- sequential, lane 0.  (The 300 Hz low-pass of :269-270
    // only feeds lpf_energy, read solely behind `chan->options & (1LL<1)`, i.e. never: it has no observable effect.)
    CHZ_WAVE_SYNC();
    int tmute = 0; double tdev = 0.0;
    if (lane == 0) {
      FmTone g{ext->g_s0, ext->g_s1, ext->old_pl_phase, ext->tone_deviation, ext->pl_sample_count, ext->tone_mute};
      const int isamprate = (int)samprate;
      const int integrate = (int)rint(isamprate * 0.24);
      for (int n = 0; n < N; n++) fm_tone_sample(g, c, esh[n], isamprate, integrate);
      ext->g_s0 = g.s0; ext->g_s1 = g.s1; ext->old_pl_phase = g.old_phase; ext->pl_sample_count = g.count;
      ext->tone_mute = g.tmute; ext->tone_deviation = g.tdev;
      tmute = g.tmute; tdev = g.tdev;
    }
    tmute = __shfl(tmute, 0); tdev = __shfl(tdev, 0);
    r.tone_deviation = tdev; r.tone_mute = tmute;
    if (tmute) {                                                                // :305-309: muted before de-emphasis runs
      r.frame = 1; r.mute = 1; r.output_power = 0.0; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
      if (lane == 0) { demod_publish(p, ch, r); p.state[ch] = st; }
      return;
    }
  }
  fm_deemph_output(p, c, st, r, ch, lane, esh, n0, cnt, N);
}

// The carrier PLL of the coherent modes (src/linear.c:83-153, loop src/osc.c:75-205) at ONE CHANNEL PER LANE.  The loop is a
// recurrence through a non-linear phase detector -- ~1.3 us of dependent double-precision arithmetic per sample -- so a
// wavefront that walks one channel keeps 1 lane of 64 busy (round 2: 330 us per 1024 channels, 0.5 M channels per 20 ms).  Here
// 64 channels' blocks are transposed through LDS, tile by tile, and the 64 lanes run their loops side by side; the mixed
// blocks go back to memory (p.mix) the same way, the loop's results to DemodExt, and demod_linear_tail picks both up.
// Statement for statement the loop of demod_linear_tail's lane-0 path: the results are bit-identical.
#define PLL_TILE 16
__global__ void __launch_bounds__(64, CHZ_PLL_WAVES) pll_lanes(DemodParams p) {
  HIP_DYNAMIC_SHARED(float2, tile)                         // [64][PLL_TILE + 1]
  const int lane = (int)threadIdx.x;
  const int base = p.ch0 + (int)blockIdx.x * 64;           // first channel of this workgroup
  const int ch = base + lane;
  const int N = p.olen;
  const bool active = (int)blockIdx.x * 64 + lane < p.nch && p.chan[ch].on && p.chan[ch].kind == 0 && p.chan[ch].pll_enable != 0;
  const unsigned long long act = __ballot(active);
  if (act == 0ull) return;                                 // wave-uniform
  PllState q; DemodExt* __restrict__ ext = p.ext + ch;
  int isamprate = 1, lock_limit = 0; bool square = false;                       // (only the few members the loop reads: the record is 200 bytes)
  double signal = 0.0, noise = 0.0, pll_foff = 0.0, sq_open = 0.0, sq_close = 0.0;
  if (active) {
    const DemodChan* __restrict__ c = p.chan + ch;
    q = ext->pll;
    isamprate = (int)c->samprate; square = c->pll_square != 0; sq_open = c->squelch_open; sq_close = c->squelch_close;
    lock_limit = (int)rint(0.5 * isamprate);                                  // DEFAULT_PLL_LOCKTIME (src/linear.c:6,38-40)
    double bw = c->pll_loop_bw / isamprate;
    if (q.lock) bw *= 0.1;
    pll_set_params(q, bw, M_SQRT1_2);                                          // DEFAULT_PLL_DAMPING (:5)
    pll_foff = ext->foffset;
  }
  constexpr int LD = PLL_TILE + 1;
  // in: row r = channel base + r, PLL_TILE consecutive samples per row, 64 / PLL_TILE rows per wavefront load.  A tile travels
  // global -> registers -> LDS with all its loads issued back to back, and the NEXT tile's loads are issued before this one is
  // walked (a load -> store round trip per row step otherwise: the wavefront then sits out the memory latency 240 times per block).
  constexpr int RPS = 64 / PLL_TILE, STEPS = 64 / RPS;
  float2 regs[STEPS];
  auto fetch_tile = [&](int t0) {
    const int tn = N - t0 < PLL_TILE ? N - t0 : PLL_TILE;
    const int n = lane % PLL_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
      const int r = k * RPS + lane / PLL_TILE;
      regs[k] = make_float2(0.f, 0.f);
      if (((act >> r) & 1ull) && n < tn) regs[k] = p.in[(size_t)(base + r) * N + t0 + n];
    }
  };
  // the AGC's first look at the block demod_linear() takes after the PLL (src/linear.c:177-203): the largest energy of a 2 ms slice of
  // the MIXED block, slices in order -- this lane walks exactly those samples in exactly that order, so it keeps the sum (round 4)
  int sps = (int)rint(N * .002 / p.blocktime);
  sps = sps < 1 ? 1 : sps;
  double peak = 0.0, energy = 0.0; int in_slice = 0;
  fetch_tile(0);
  for (int t0 = 0; t0 < N; t0 += PLL_TILE) {
    const int tn = N - t0 < PLL_TILE ? N - t0 : PLL_TILE;
    {
      const int n = lane % PLL_TILE;
#pragma unroll
      for (int k = 0; k < STEPS; k++) tile[(k * RPS + lane / PLL_TILE) * LD + n] = regs[k];
    }
    CHZ_WAVE_SYNC();
    if (t0 + PLL_TILE < N) fetch_tile(t0 + PLL_TILE);
    if (active) {
      // one sample of the loop (src/linear.c:83-153 with the loop filter of src/osc.c:75-205), in the reference's order
      auto pll_step = [&](const float2 v, const int n) -> float2 {
        double sn, cs; pll_nco(q.vco_phase, sn, cs);
        const double br = v.x, bi = v.y;
        const double sr = br * cs + bi * sn, si = bi * cs - br * sn;           // buffer[n] * conj(vco)
        const float2 mixed = make_float2((float)sr, (float)si);
        {
          float a = mixed.x * mixed.x, b = mixed.y * mixed.y;
          CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
          energy += (double)(a + b);                                          // cnrmf, as demod_lin_lanes' first pass summed it
          if (++in_slice == sps) {
            if (t0 + n + 1 < N && energy > peak) peak = energy;
            energy = 0.0; in_slice = 0;
          }
        }
        double phase;
        if (q.lock) {
          if (!square) { const double mag = sqrt(sr * sr + si * si); phase = (mag > 0) ? si / mag : 0.0; }
          else phase = sr * si / (sr * sr - si * si);
        } else {
          if (!square) phase = atan2(si, sr);
          else phase = 0.5 * atan2(sr * si + si * sr, sr * sr - si * si);      // carg(s*s)
        }
        phase /= (2 * M_PI);
        pll_foff = isamprate * pll_run(q, phase);
        signal += sr * sr; noise += si * si;
        return mixed;
      };
      int n = 0;
#if CHZ_PLL_UNROLL > 1
      for (; n + CHZ_PLL_UNROLL <= tn; n += CHZ_PLL_UNROLL) {      // (see demod_lin_lanes: one LDS round trip per group instead of per sample)
        float2 v[CHZ_PLL_UNROLL];
#pragma unroll
        for (int u = 0; u < CHZ_PLL_UNROLL; u++) v[u] = tile[lane * LD + n + u];
#pragma unroll
        for (int u = 0; u < CHZ_PLL_UNROLL; u++) v[u] = pll_step(v[u], n + u);
#pragma unroll
        for (int u = 0; u < CHZ_PLL_UNROLL; u++) tile[lane * LD + n + u] = v[u];
      }
#endif
      for (; n < tn; n++) tile[lane * LD + n] = pll_step(tile[lane * LD + n], n);
    }
    CHZ_WAVE_SYNC();
    for (int r0 = 0; r0 < 64; r0 += 64 / PLL_TILE) {
      const int r = r0 + lane / PLL_TILE, n = lane % PLL_TILE;
      if (((act >> r) & 1ull) && n < tn) p.mix[(size_t)(base + r) * N + t0 + n] = tile[r * LD + n];
    }
    CHZ_WAVE_SYNC();
  }
  if (active) {
    double pll_snr;
    const double pll_cph = ldexp(2 * M_PI * (double)q.vco_phase, -32);
    int pll_rot = q.wraps;
    if (noise != 0) { pll_snr = (signal / noise) - 1; if (pll_snr < 0) pll_snr = 0; }
    else pll_snr = __builtin_nan("");
    if (pll_snr < sq_close) {
      q.lock_count -= N;
      if (q.lock_count <= -lock_limit) { q.lock_count = -lock_limit; q.lock = 0; }
    } else if (pll_snr > sq_open) {
      q.lock_count += N;
      if (q.lock_count >= lock_limit) {
        q.lock_count = lock_limit;
        if (!q.lock) { q.lock = 1; pll_rot = 0; }
      }
    }
    ext->pll = q; ext->pll_snr = pll_snr; ext->pll_cphase = pll_cph; ext->foffset = pll_foff; ext->pll_rotations = pll_rot;
    if (p.agc_peak != nullptr && p.peak_pll != 0) p.agc_peak[ch] = peak;
  }
}

// ---- FM's two sample-by-sample recurrences at ONE CHANNEL PER LANE (src/fm.c:176-203 the PLL demodulator, :264-311 the PL-tone
// detector).  Inside demod_fm_wave lane 0 walks the block while 63 lanes wait: 66 ns per channel with the PLL demodulator and 6.1 ns
// with the tone detector, against 2.4 ns for plain FM (profiles/r03_fm_probe.jsonl).  demod_fm() is therefore cut where the loops sit:
//   fm_front_k      (a wavefront per channel)  noise smoothing, SNR, squelch sequencer          -> ext.fm_go / fm_snr / fm_noise, state
//   fm_pll_lanes    (a LANE per channel)       the PLL loop over the block                      -> mix[0..N) baseband, ext.pll
//   demod_linear_tail                          everything between the loops                     -> mix[N..2N) tone input, ext.fm_stage
//   fm_tone_lanes   (a LANE per channel)       the Goertzel detector and its decision           -> ext tone state
//   fm_finish       (a wavefront per channel)  mute, or de-emphasis + gain + PCM + status
// Only channels that use the PLL demodulator take the first two, only channels with a tone squelch the last two; the statements are
// the ones demod_fm_wave runs (fm_front, fm_pll_sample, fm_tone_sample, fm_deemph_output), so the frames are bit-identical.
__global__ void __launch_bounds__(64) fm_front_k(DemodParams p) {
  HIP_DYNAMIC_SHARED(double, esh)                          // [N]
  const int lane = (int)threadIdx.x;
  const int ch = p.ch0 + (int)blockIdx.x;
  const DemodChan* __restrict__ cp = p.chan + ch;
  if (!cp->on || cp->kind != 1 || cp->pll_enable == 0) return;                // wave-uniform
  const DemodChan c = *cp;
  DemodState st = p.state[ch];
  const int N = p.olen, SEG = (N + 63) >> 6, n0 = lane * SEG;
  const int cnt = n0 >= N ? 0 : (N - n0 < SEG ? N - n0 : SEG);
  const FmFront f = fm_front(p, c, st, ch, p.in + (size_t)ch * N, n0, cnt, N, esh);
  if (lane == 0) {
    DemodExt* __restrict__ ext = p.ext + ch;
    ext->fm_snr = f.fmsnr; ext->fm_noise = f.noise; ext->fm_go = st.squelch_state > 4;
    p.state[ch] = st;
  }
}

#define FM_TILE 32
#define FMP_TILE 16
__global__ void __launch_bounds__(64, 2) fm_pll_lanes(DemodParams p) {
  HIP_DYNAMIC_SHARED(float2, tile)                         // [64][FMP_TILE + 1]
  const int lane = (int)threadIdx.x;
  const int base = p.ch0 + (int)blockIdx.x * 64;
  const int ch = base + lane;
  const int N = p.olen;
  DemodExt* __restrict__ ext = p.ext + ch;
  const bool active = (int)blockIdx.x * 64 + lane < p.nch && p.chan[ch].on && p.chan[ch].kind == 1 && p.chan[ch].pll_enable != 0 && ext->fm_go != 0;
  const unsigned long long act = __ballot(active);
  if (act == 0ull) return;                                 // wave-uniform
  PllState q; double pdev = 0.0, noise = 0.0; bool extend = false;
  if (active) {
    const DemodChan* __restrict__ c = p.chan + ch;
    q = ext->pll;
    const int isamprate = (int)c->samprate;
    pdev = 5000.0 / isamprate; noise = ext->fm_noise; extend = c->threshold_extend != 0;
    if (!p.state[ch].pll_was_on) {                         // (demod_linear_tail sets it behind us)
      pll_init(q);
      pll_set_params(q, 500.0 / isamprate, M_SQRT1_2);
      q.lower = -pdev; q.upper = +pdev;
    }
  }
  constexpr int LD = FMP_TILE + 1;
  float* __restrict__ mixf = reinterpret_cast<float*>(p.mix);
  for (int t0 = 0; t0 < N; t0 += FMP_TILE) {
    const int tn = N - t0 < FMP_TILE ? N - t0 : FMP_TILE;
    {   // all of the tile's loads in flight at once, then into LDS (the loop's registers leave no room to hold the next tile as well)
      constexpr int RPS = 64 / FMP_TILE, STEPS = 64 / RPS;
      float2 regs[STEPS];
      const int n = lane % FMP_TILE;
#pragma unroll
      for (int k = 0; k < STEPS; k++) {
        const int r = k * RPS + lane / FMP_TILE;
        regs[k] = make_float2(0.f, 0.f);
        if (((act >> r) & 1ull) && n < tn) regs[k] = p.in[(size_t)(base + r) * N + t0 + n];
      }
#pragma unroll
      for (int k = 0; k < STEPS; k++) tile[(k * RPS + lane / FMP_TILE) * LD + n] = regs[k];
    }
    CHZ_WAVE_SYNC();
    if (active)
      for (int n = 0; n < tn; n++) tile[lane * LD + n].x = fm_pll_sample(q, tile[lane * LD + n], pdev, extend, 0.5, noise);
    CHZ_WAVE_SYNC();
    for (int r0 = 0; r0 < 64; r0 += 64 / FMP_TILE) {
      const int r = r0 + lane / FMP_TILE, n = lane % FMP_TILE;
      if (((act >> r) & 1ull) && n < tn) mixf[(size_t)(base + r) * 2 * N + t0 + n] = tile[r * LD + n].x;
    }
    CHZ_WAVE_SYNC();
  }
  if (active) ext->pll = q;
}

__global__ void __launch_bounds__(64, 2) fm_tone_lanes(DemodParams p) {
  HIP_DYNAMIC_SHARED(float, tilef)                         // [64][FM_TILE + 1]
  const int lane = (int)threadIdx.x;
  const int base = p.ch0 + (int)blockIdx.x * 64;
  const int ch = base + lane;
  const int N = p.olen;
  DemodExt* __restrict__ ext = p.ext + ch;
  const bool active = (int)blockIdx.x * 64 + lane < p.nch && p.chan[ch].on && p.chan[ch].kind == 1 && p.chan[ch].tone_freq != 0 && ext->fm_stage == 1;
  const unsigned long long act = __ballot(active);
  if (act == 0ull) return;                                 // wave-uniform
  FmTone g{0.0, 0.0, 0.0, 0.0, 0, 0};
  DemodChan c; c.g_coeff = 0.0; c.g_cfr = 0.0; c.g_cfi = 0.0; c.tone_freq = 0.0;      // (the four members fm_tone_sample reads)
  int isamprate = 1, integrate = 1;
  if (active) {
    const DemodChan* __restrict__ cp = p.chan + ch;
    c.g_coeff = cp->g_coeff; c.g_cfr = cp->g_cfr; c.g_cfi = cp->g_cfi; c.tone_freq = cp->tone_freq;
    isamprate = (int)cp->samprate; integrate = (int)rint(isamprate * 0.24);
    g = FmTone{ext->g_s0, ext->g_s1, ext->old_pl_phase, ext->tone_deviation, ext->pl_sample_count, ext->tone_mute};
  }
  constexpr int LD = FM_TILE + 1;
  const float* __restrict__ mixf = reinterpret_cast<const float*>(p.mix);
  // (tiles staged through registers, the next one fetched while this one is walked: see pll_lanes)
  constexpr int RPS = 64 / FM_TILE, STEPS = 64 / RPS;
  float regs[STEPS];
  auto fetch_tile = [&](int t0) {
    const int tn = N - t0 < FM_TILE ? N - t0 : FM_TILE;
    const int n = lane % FM_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
      const int r = k * RPS + lane / FM_TILE;
      regs[k] = 0.f;
      if (((act >> r) & 1ull) && n < tn) regs[k] = mixf[(size_t)(base + r) * 2 * N + N + t0 + n];
    }
  };
  fetch_tile(0);
  for (int t0 = 0; t0 < N; t0 += FM_TILE) {
    const int tn = N - t0 < FM_TILE ? N - t0 : FM_TILE;
    {
      const int n = lane % FM_TILE;
#pragma unroll
      for (int k = 0; k < STEPS; k++) tilef[(k * RPS + lane / FM_TILE) * LD + n] = regs[k];
    }
    CHZ_WAVE_SYNC();
    if (t0 + FM_TILE < N) fetch_tile(t0 + FM_TILE);
    if (active)
      for (int n = 0; n < tn; n++) fm_tone_sample(g, c, (double)tilef[lane * LD + n], isamprate, integrate);
    CHZ_WAVE_SYNC();
  }
  if (active) {
    ext->g_s0 = g.s0; ext->g_s1 = g.s1; ext->old_pl_phase = g.old_phase; ext->pl_sample_count = g.count;
    ext->tone_mute = g.tmute; ext->tone_deviation = g.tdev;
  }
}

__global__ void __launch_bounds__(64) fm_finish(DemodParams p) {
  HIP_DYNAMIC_SHARED(double, esh)                          // [N]
  const int lane = (int)threadIdx.x;
  const int ch = p.ch0 + (int)blockIdx.x;
  const DemodChan* __restrict__ cp = p.chan + ch;
  DemodExt* __restrict__ ext = p.ext + ch;
  if (!cp->on || cp->kind != 1 || cp->tone_freq == 0 || ext->fm_stage != 1) return;     // wave-uniform
  const DemodChan c = *cp;
  DemodState st = p.state[ch];
  const int N = p.olen, SEG = (N + 63) >> 6, n0 = lane * SEG;
  const int cnt = n0 >= N ? 0 : (N - n0 < SEG ? N - n0 : SEG);
  const float* __restrict__ mixf = reinterpret_cast<const float*>(p.mix) + (size_t)ch * 2 * N + N;
  for (int i = 0; i < cnt; i++) esh[n0 + i] = (double)mixf[n0 + i];
  DemodStatus r;
  r.pll_lock = 0; r.pll_snr = 0.0; r.pll_cphase = 0.0; r.pll_rotations = 0;
  r.n0 = st.n0; r.snr = ext->fm_snr; r.squelch_state = st.squelch_state; r.gain = 0.0;
  r.tone_deviation = ext->tone_deviation; r.tone_mute = ext->tone_mute;
  CHZ_WAVE_SYNC();                                         // every lane has read the record before lane 0 marks it done
  if (lane == 0) ext->fm_stage = 0;
  if (r.tone_mute) {                                       // :305-309: muted before de-emphasis runs
    r.frame = 1; r.mute = 1; r.output_power = 0.0; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
    if (lane == 0) { demod_publish(p, ch, r); p.state[ch] = st; }
    return;
  }
  fm_deemph_output(p, c, st, r, ch, lane, esh, n0, cnt, N);
}

// ---- demod_linear() at ONE CHANNEL PER LANE.  A wavefront per channel spends ~800 vector instructions per channel and block, most
// of them on what is the same in all 64 lanes (noise smoothing, the AGC's decisions with their square roots, divisions and pow(), the
// squelch, the status record), for 3.75 samples per lane: 2.2 ns per channel at 1.5 M channels, the slowest stage of the chain.  Here a
// workgroup of 64 lanes takes 64 channels; their blocks pass through LDS in tiles of 32 samples (coalesced rows in, one column per
// lane out, as in pll_lanes), every lane walks ITS channel in the reference's own order -- slice energies and peak (src/linear.c:
// 177-234), the gain ramp multiplied up sample by sample, the carrier filter as the recurrence it is, the power sum in sample order --
// so nothing is re-associated: statement for statement chzo_lindemod_block / demod_linear().  The samples of the final pass go back
// into the tile where the input stood and leave as packed PCM, whole rows at a time.  Serves every channel of the linear demodulator
// (those in a coherent mode after pll_lanes has mixed their block down); demod_linear_tail then only sees the FM channels.
#define LIN_TILE 16
struct LinRow { unsigned char* o; int enc, channels, data; };
__global__ void __launch_bounds__(64, CHZ_LIN_WAVES) demod_lin_lanes(DemodParams p) {
  HIP_DYNAMIC_SHARED(float2, tile)                         // [64][LIN_TILE + 1], then LinRow[64]
  constexpr int LD = LIN_TILE + 1;
  LinRow* rows = reinterpret_cast<LinRow*>(tile + 64 * LD);
  const int lane = (int)threadIdx.x;
  const int base = p.ch0 + (int)blockIdx.x * 64;
  const int ch = base + lane;
  const int N = p.olen;
  bool active = false, pll = false;
  if ((int)blockIdx.x * 64 + lane < p.nch) {
    const DemodChan* __restrict__ cp = p.chan + ch;
    pll = cp->pll_enable != 0;
    active = cp->on && cp->kind == 0 && (!pll || (p.mix != nullptr && p.lin_pll != 0));
  }
  const unsigned long long act = __ballot(active);
  if (act == 0ull) return;                                 // wave-uniform
  const unsigned long long from_mix = __ballot(active && pll);      // rows whose block pll_lanes left in p.mix
  // the channel's record, member by member (200 bytes per lane otherwise)
  int channels = 1, env = 0, agc = 0, snr_squelch = 0, squelch_tail = 0, tuned = 1, enc = 0;
  double samprate = 1.0, headroom = 0.0, threshold = 0.0, hangtime = 0.0, dc_alpha = 0.0, bandwidth = 1.0, sq_open = 0.0, sq_close = 0.0;
  double osc_phase0 = 0.0, osc_freq = 0.0, recov_ps = 1.0;
  unsigned osc_job0 = 0;
  DemodState st; st.gain = 0.0; st.am_dc = 0.0; st.n0 = 0.0; st.hangcount = 0; st.squelch_state = 0; st.squelch_open = 0;
  double bb_power = 0.0;
  double pll_snr = 0.0, pll_cph = 0.0, pll_foff = 0.0; int pll_lock = 0, pll_rot = 0;
  if (active) {
    const DemodChan* __restrict__ c = p.chan + ch;
    channels = c->channels; env = c->env; agc = c->agc; snr_squelch = c->snr_squelch; squelch_tail = c->squelch_tail; tuned = c->tuned; enc = c->encoding;
    samprate = c->samprate; headroom = c->headroom; threshold = c->threshold; hangtime = c->hangtime; dc_alpha = c->dc_alpha;
    bandwidth = c->bandwidth; sq_open = c->squelch_open; sq_close = c->squelch_close;
    osc_phase0 = c->osc_phase0; osc_freq = c->osc_freq; osc_job0 = c->osc_job0; recov_ps = c->recov_ps;
    st = p.state[ch];
    bb_power = p.power[ch];
    const double est = p.n0[ch];                           // src/radio.c:1466-1473
    if (st.n0 != st.n0) st.n0 = est;
    else { const double diff = est - st.n0; st.n0 += p.power_alpha * diff; }
    if (pll) {
      const DemodExt* __restrict__ ext = p.ext + ch;
      pll_snr = ext->pll_snr; pll_cph = ext->pll_cphase; pll_foff = ext->foffset; pll_lock = ext->pll.lock; pll_rot = ext->pll_rotations;
    }
  }
  // chan->shift (src/linear.c:168-172): the phasor at a tile's first sample from the closed form, stepped in double inside the tile
  const bool rot = active && osc_freq != 0.0;
  double c1 = 1.0, s1 = 0.0;
  if (rot) sincospi(2.0 * osc_freq, &s1, &c1);
  auto rot_at = [&](int n, double& cr, double& sr) {
    const double g = (double)(p.job - osc_job0) * (double)N + (double)n;
    double hi = g * osc_freq, lo = fma(g, osc_freq, -hi);
    hi -= rint(hi);
    sincospi(2.0 * (osc_phase0 + hi + lo), &sr, &cr);
  };
  // A tile travels global -> registers -> LDS: all of a tile's loads are issued back to back, and the NEXT tile's are issued before
  // this one is worked on (32 dependent load -> store round trips per tile otherwise: the kernel then waits for memory 16 times per block)
  constexpr int ROWS_PER_STEP = 64 / LIN_TILE, STEPS = 64 / ROWS_PER_STEP;
  float2 regs[STEPS];
  auto fetch_tile = [&](int t0) {
    const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
    const int n = lane % LIN_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
      const int r = k * ROWS_PER_STEP + lane / LIN_TILE;
      regs[k] = make_float2(0.f, 0.f);
      if (((act >> r) & 1ull) && n < tn) {
        const float2* __restrict__ src = ((from_mix >> r) & 1ull) ? p.mix : p.in;
        regs[k] = src[(size_t)(base + r) * N + t0 + n];
      }
    }
  };
  auto place_tile = [&]() {
    const int n = lane % LIN_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) tile[(k * ROWS_PER_STEP + lane / LIN_TILE) * LD + n] = regs[k];
  };
  // ---- AGC (src/linear.c:177-234): the largest slice energy of the block, slices in order
  double gain_change = 1.0;
  // whoever had the block's samples in hand last has already looked at it -- the channel kernel (rows in LDS), or pll_lanes for a
  // coherent-mode channel (its lane walks the mixed block in order anyway): the first pass over the baseband is only walked by lanes
  // with a post-detection shift, which rotates the block once more
  const bool have_peak = p.agc_peak != nullptr && !rot && (pll ? p.peak_pll != 0 : p.peak_chan != 0);
  const bool walk1 = active && agc && !have_peak;
  if (__ballot(active && agc) != 0ull) {                   // wave-uniform
    int sps = (int)rint(N * .002 / p.blocktime);
    sps = sps < 1 ? 1 : sps;
    double peak = 0.0, energy = 0.0; int in_slice = 0;
    const bool any_walk = __ballot(walk1) != 0ull;         // wave-uniform
    if (any_walk) fetch_tile(0);
    for (int t0 = 0; any_walk && t0 < N; t0 += LIN_TILE) {
      const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
      place_tile();
      CHZ_WAVE_SYNC();
      if (t0 + LIN_TILE < N) fetch_tile(t0 + LIN_TILE);
      if (walk1) {
        double cr = 1.0, sr = 0.0;
        if (rot) rot_at(t0, cr, sr);
        for (int n = 0; n < tn; n++) {
          float2 v = tile[lane * LD + n];
          if (rot) {
            const double xr = v.x, xi = v.y;
            v = make_float2((float)(xr * cr - xi * sr), (float)(xr * sr + xi * cr));
            const double nc = cr * c1 - sr * s1; sr = cr * s1 + sr * c1; cr = nc;
          }
          float a = v.x * v.x, b = v.y * v.y;
          CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
          energy += (double)(a + b);                       // cnrmf
          if (++in_slice == sps) {
            // `while (n + samples_per_slice < N)` (:199): a slice counts only if it ends before the block's last sample
            if (t0 + n + 1 < N && energy > peak) peak = energy;
            energy = 0.0; in_slice = 0;
          }
        }
      }
      CHZ_WAVE_SYNC();
    }
    if (active && agc) {
      if (have_peak) peak = p.agc_peak[ch];
      const double bn = sqrt(bandwidth * st.n0);
      const double ampl = sqrt(bb_power);
      const double peak_level = sqrt(peak / sps);
      if (peak_level * st.gain > M_SQRT2 * headroom) {
        st.gain = M_SQRT2 * headroom / peak_level;
        gain_change = 1.0;
        st.hangcount = (int)rint(0.08 * samprate);
      } else if (ampl * st.gain > headroom) {
        const double newgain = headroom / ampl;
        if (newgain > 0) gain_change = pow(newgain / st.gain, 1.0 / N);
        st.hangcount = (int)rint(hangtime * samprate);
      } else if (bn * st.gain > threshold * headroom) {
        const double newgain = threshold * headroom / bn;
        if (newgain > 0) gain_change = pow(newgain / st.gain, 1.0 / N);
      } else if (st.hangcount > 0) {
        st.hangcount -= N;
      } else {
        gain_change = recov_ps;                            // pow(recovery_rate, 1 / samprate)
      }
    }
  }
  // ---- squelch sequencer (src/linear.c:313-352); it does not interact with the final pass, and knowing the frame type first
  // saves packing PCM nobody will send
  double snr = __builtin_huge_val();
  if (snr_squelch) snr = (bb_power / (st.n0 * bandwidth)) - 1.0;
  else if (pll) snr = pll_snr;                             // :317-318
  const int smax = squelch_tail + 4;
  if (!(snr_squelch || pll) || snr >= sq_open) st.squelch_state = smax;
  else if (st.squelch_state > 0 && snr < sq_close) st.squelch_state--;
  const bool data = active && st.squelch_state >= 4;
  rows[lane] = LinRow{p.pcm + (size_t)ch * p.pcm_stride, enc, channels, data ? 1 : 0};
  // ---- final pass (src/linear.c:236-311); the gain ramp and the carrier filter run whether or not the frame is sent
  const double k_env = M_SQRT1_2;
  const bool dcfilt = env && dc_alpha != 0;
  double gain = st.gain, am_dc = st.am_dc, part = 0.0;
  const unsigned long long any_data = __ballot(data);
  // every row that sends is mono S16 and the PCM rows are 8-byte aligned (wave-uniform): the packed store below
  const bool s16_mono = CHZ_LIN_PACKED_STORE && (p.pcm_stride & 7) == 0 &&
                        __ballot(data && !(channels == 1 && (enc == CHZ_PCM_S16BE_K || enc == CHZ_PCM_S16LE_K))) == 0ull;
  fetch_tile(0);
  for (int t0 = 0; t0 < N; t0 += LIN_TILE) {
    const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
    place_tile();
    CHZ_WAVE_SYNC();
    if (t0 + LIN_TILE < N) fetch_tile(t0 + LIN_TILE);
    if (active) {
      double cr = 1.0, sr = 0.0;
      if (rot) rot_at(t0, cr, sr);
      // one sample of the final pass, in the reference's order (the state -- gain, carrier filter, power sum, shift phasor -- lives in the captured variables)
      auto final_step = [&](float2 v) -> float2 {
        if (rot) {
          const double xr = v.x, xi = v.y;
          v = make_float2((float)(xr * cr - xi * sr), (float)(xr * sr + xi * cr));
          const double nc = cr * c1 - sr * s1; sr = cr * s1 + sr * c1; cr = nc;
        }
        float oa, ob = 0.f;
        if (channels == 1) {
          double sgn;
          if (env) {
            sgn = gain * k_env * (double)demod_cabsf(v);
            gain *= gain_change;
            part += sgn * sgn;
            if (dcfilt) { am_dc += dc_alpha * (sgn - am_dc); sgn -= am_dc; }
          } else {
            sgn = gain * (double)v.x;
            gain *= gain_change;
            part += sgn * sgn;
          }
          oa = (float)sgn;
        } else {
          double a, b;
          if (env) {
            const double k = gain * k_env;
            a = k * (double)v.x; b = k * (double)demod_cabsf(v);
            gain *= gain_change;
            part += a * a + b * b;
            if (dcfilt) { am_dc += dc_alpha * (b - am_dc); b -= am_dc; }
          } else {
            a = gain * (double)v.x; b = gain * (double)v.y;
            gain *= gain_change;
            part += a * a + b * b;
          }
          oa = (float)a; ob = (float)b;
        }
        return make_float2(oa, ob);
      };
      int n = 0;
#if CHZ_LIN_UNROLL > 1
      // the walk is a chain of LDS round trips (read a sample, a few dependent double-precision operations, write it back): CHZ_LIN_UNROLL
      // samples are read together, stepped in order, and written together -- the same operations in the same order, one LDS latency per group
      for (; n + CHZ_LIN_UNROLL <= tn; n += CHZ_LIN_UNROLL) {
        float2 v[CHZ_LIN_UNROLL];
#pragma unroll
        for (int u = 0; u < CHZ_LIN_UNROLL; u++) v[u] = tile[lane * LD + n + u];
#pragma unroll
        for (int u = 0; u < CHZ_LIN_UNROLL; u++) v[u] = final_step(v[u]);
#pragma unroll
        for (int u = 0; u < CHZ_LIN_UNROLL; u++) tile[lane * LD + n + u] = v[u];
      }
#endif
      for (; n < tn; n++) tile[lane * LD + n] = final_step(tile[lane * LD + n]);
    }
    CHZ_WAVE_SYNC();
    if (any_data != 0ull) {
      if (s16_mono && tn == LIN_TILE) {
        // mono S16 (what a voice channel sends): a lane packs FOUR samples of a row and stores them as one 8-byte word -- 4 store
        // instructions per tile, 16 rows each, instead of 16 of 4 rows with 2 bytes per lane; the values are demod_put's
        static_assert(LIN_TILE == 16, "the packed store walks a tile as 4 x 4 samples");
        for (int r0 = 0; r0 < 64; r0 += 16) {
          const int r = r0 + (lane >> 2), n4 = (lane & 3) * 4;
          if ((any_data >> r) & 1ull) {
            const LinRow q = rows[r];
            const bool be = q.enc == CHZ_PCM_S16BE_K;
            const unsigned long long w = (unsigned long long)demod_s16(tile[r * LD + n4].x, be) | ((unsigned long long)demod_s16(tile[r * LD + n4 + 1].x, be) << 16) |
                                         ((unsigned long long)demod_s16(tile[r * LD + n4 + 2].x, be) << 32) | ((unsigned long long)demod_s16(tile[r * LD + n4 + 3].x, be) << 48);
            *reinterpret_cast<unsigned long long*>(q.o + 2 * (t0 + n4)) = w;
          }
        }
      } else
      for (int r0 = 0; r0 < 64; r0 += 64 / LIN_TILE) {
        const int r = r0 + lane / LIN_TILE, n = lane % LIN_TILE;
        if (((any_data >> r) & 1ull) && n < tn) {
          const LinRow q = rows[r];
          const float2 v = tile[r * LD + n];
          if (q.channels == 1) demod_put(q.o, q.enc, t0 + n, v.x);
          else { demod_put(q.o, q.enc, 2 * (t0 + n), v.x); demod_put(q.o, q.enc, 2 * (t0 + n) + 1, v.y); }
        }
      }
    }
    CHZ_WAVE_SYNC();
  }
  if (!active) return;
  st.gain = gain; st.am_dc = am_dc;
  double output_power = part / N;
  if (channels == 1) output_power *= 2;
  DemodStatus r;
  r.gain = st.gain; r.n0 = st.n0; r.snr = snr; r.squelch_state = st.squelch_state;
  r.output_power = output_power; r.foffset = pll_foff; r.pdeviation = 0.0;
  r.pll_lock = pll_lock; r.pll_snr = pll_snr; r.pll_cphase = pll_cph; r.pll_rotations = pll_rot; r.tone_deviation = 0.0; r.tone_mute = 0;
  if (!data) {
    r.frame = 1; r.mute = st.squelch_state == 0;
    if (st.squelch_state == 3 || st.squelch_state == 0) r.output_power = 0;
  } else {
    if (snr_squelch || pll) {
      if (snr < sq_close) st.squelch_open = 0;
      else if (!st.squelch_open && snr > sq_open) { st.squelch_open = 1; st.am_dc = 0; }
    } else st.squelch_open = 1;
    r.frame = 0;
    r.mute = (output_power == 0 || !st.squelch_open || !tuned);
  }
  demod_publish(p, ch, r);
  // (only the members this demodulator owns: the FM members of the record are not ours to touch)
  DemodState* __restrict__ so = p.state + ch;
  so->gain = st.gain; so->am_dc = st.am_dc; so->n0 = st.n0; so->hangcount = st.hangcount; so->squelch_state = st.squelch_state; so->squelch_open = st.squelch_open;
}

// ---- demod_fm() at ONE CHANNEL PER LANE (channels without the PLL demodulator; those keep the passes above).  With the squelch open
// radiod's default SNR estimator is the amplitude-variance one (src/fm.c:110-129): two sums over the block and fm_snr()'s Bessel
// iteration, per channel, the same in all 64 lanes of a wavefront-per-channel kernel -- 4.9 ns per channel at 1.5 M channels against
// 2.4 with the bb_power / N0 estimator.  Here each lane walks its own channel through demod_fm() in the reference's order, the blocks
// passing through LDS in 16-sample tiles exactly as in demod_lin_lanes:
//   pass 1, 2   mean amplitude, amplitude variance              (only when a lane's estimator needs them)
//   pass 3      discriminator with threshold extension -> baseband (floats, into the channel's `mix` block), offset / deviation sums
//   pass 4      PL-tone detector over the baseband after DC removal  (only when a lane has a tone squelch)
//   pass 5      de-emphasis as the recurrence it is, gain, PCM
// Nothing is re-associated: statement for statement chzo_fmdemod_block / demod_fm().
__global__ void __launch_bounds__(64, CHZ_FM_WAVES) demod_fm_lanes(DemodParams p) {
  HIP_DYNAMIC_SHARED(float2, tile)                         // [64][LIN_TILE + 1] float2 (passes 1-3) / floats (4, 5); then LinRow[64]
  constexpr int LD = LIN_TILE + 1;
  float* tilef = reinterpret_cast<float*>(tile);
  LinRow* rows = reinterpret_cast<LinRow*>(tile + 64 * LD);
  const int lane = (int)threadIdx.x;
  const int base = p.ch0 + (int)blockIdx.x * 64;
  const int ch = base + lane;
  const int N = p.olen;
  bool active = false;
  if ((int)blockIdx.x * 64 + lane < p.nch) {
    const DemodChan* __restrict__ cp = p.chan + ch;
    active = cp->on && cp->kind == 1 && cp->pll_enable == 0;
  }
  const unsigned long long act = __ballot(active);
  if (act == 0ull) return;                                 // wave-uniform
  int snr_squelch = 0, squelch_tail = 0, enc = 0; bool extend = false, tone = false;
  double samprate = 1.0, headroom = 0.0, bandwidth = 1.0, sq_open = 0.0, sq_close = 0.0, deemph_rate = 0.0, deemph_gain = 0.0;
  DemodChan ct; ct.g_coeff = 0.0; ct.g_cfr = 0.0; ct.g_cfi = 0.0; ct.tone_freq = 0.0;       // (the four members fm_tone_sample reads)
  DemodState st; st.n0 = 0.0; st.squelch_state = 0; st.pm_re = 0.0; st.pm_im = 0.0; st.deemph_state = 0.0; st.foffset = 0.0; st.pdeviation = 0.0;
  st.gain = 0.0; st.am_dc = 0.0; st.hangcount = 0; st.squelch_open = 0; st.pll_was_on = 0;
  double bb_power = 0.0;
  if (active) {
    const DemodChan* __restrict__ c = p.chan + ch;
    snr_squelch = c->snr_squelch; squelch_tail = c->squelch_tail; enc = c->encoding; extend = c->threshold_extend != 0; tone = c->tone_freq != 0;
    samprate = c->samprate; headroom = c->headroom; bandwidth = c->bandwidth; sq_open = c->squelch_open; sq_close = c->squelch_close;
    deemph_rate = c->deemph_rate; deemph_gain = c->deemph_gain;
    if (tone) { ct.g_coeff = c->g_coeff; ct.g_cfr = c->g_cfr; ct.g_cfi = c->g_cfi; ct.tone_freq = c->tone_freq; }
    st = p.state[ch];
    bb_power = p.power[ch];
    const double est = p.n0[ch];                           // src/radio.c:1466-1473
    if (st.n0 != st.n0) st.n0 = est;
    else { const double diff = est - st.n0; st.n0 += p.power_alpha * diff; }
  }
  const double devmax = 5000.0, beta = 0.5;                // src/fm.c:43,103
  const double noise = st.n0 * bandwidth;                  // :101
  const double snr = noise == 0 ? __builtin_huge_val() : (bb_power / noise) - 1.0;
  // tiles of the channels' blocks: global -> registers -> LDS, the next tile in flight while this one is walked (see demod_lin_lanes)
  constexpr int RPS = 64 / LIN_TILE, STEPS = 64 / RPS;
  float2 regs[STEPS];
  auto fetch_x = [&](int t0, unsigned long long need) {
    const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
    const int n = lane % LIN_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
      const int r = k * RPS + lane / LIN_TILE;
      regs[k] = make_float2(0.f, 0.f);
      if (((need >> r) & 1ull) && n < tn) regs[k] = p.in[(size_t)(base + r) * N + t0 + n];
    }
  };
  auto place_x = [&]() {
    const int n = lane % LIN_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) tile[(k * RPS + lane / LIN_TILE) * LD + n] = regs[k];
  };
  float* __restrict__ mixf = reinterpret_cast<float*>(p.mix);                    // [cap][2 * N] floats: this kernel's baseband in the first N
  auto fetch_b = [&](int t0, unsigned long long need) {
    const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
    const int n = lane % LIN_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
      const int r = k * RPS + lane / LIN_TILE;
      regs[k].x = 0.f;
      if (((need >> r) & 1ull) && n < tn) regs[k].x = mixf[(size_t)(base + r) * 2 * N + t0 + n];
    }
  };
  auto place_b = [&]() {
    const int n = lane % LIN_TILE;
#pragma unroll
    for (int k = 0; k < STEPS; k++) tilef[(k * RPS + lane / LIN_TILE) * LD + n] = regs[k].x;
  };
  // the discriminator (:204-231): phase of x[n] * conj(x[n-1]); the sample before the block is phase_memory.  One sample of it:
  double psum = 0.0, pmax = 0.0, pmin = 0.0;
  double pr = st.pm_re, pi = st.pm_im;
  double p0 = pr * pr + pi * pi;                           // cnrm(phase_memory)
  if (p0 > 0) p0 /= (p0 + beta * noise);
  // (round 6) split in two: disc_phase() is the long part -- a double-precision atan2 -- and depends on nothing but the sample and its
  // predecessor; disc_commit() is the reference's sequential part (threshold extension, sums, extremes, phase_memory).  The loops below take
  // CHZ_FM_DISC_UNROLL samples at a time: their atan2 chains are independent and overlap, the commits then run in sample order -- the same
  // operations on the same values as one sample at a time (bit-identical), the dependent-latency chain per sample a fraction of it.
  auto disc_phase = [&](const float2 v, const double qr, const double qi) -> double {
    const double br = v.x, bi = v.y;
    const double sr = br * qr + bi * qi, si = bi * qr - br * qi;
    return M_1_PI * atan2(si, sr);
  };
  auto disc_commit = [&](const float2 v, double phase) -> float {
    if (extend) {
      if (fabs(phase) > devmax / samprate) phase = copysign(devmax / samprate, phase);
      float a = v.x * v.x, b = v.y * v.y; CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
      double p1 = (double)(a + b);
      if (p1 > 0) p1 /= (p1 + beta * noise);
      phase *= p0 * p1;
      p0 = p1;
    }
    const float bbf = (float)phase;
    const double bbv = (double)bbf;
    psum += bbv;
    if (bbv > pmax) pmax = bbv;
    if (bbv < pmin) pmin = bbv;
    pr = v.x; pi = v.y;
    return bbf;
  };
  auto discriminate = [&](const float2 v) -> float { return disc_commit(v, disc_phase(v, pr, pi)); };
  // ---- SNR (:105-129).  The variance estimator walks the block twice (mean amplitude, then the variance around it); the second walk
  // carries the discriminator along SPECULATIVELY (round 4): a channel that needs the estimator has its squelch open or opening, so
  // the discriminator will almost always be wanted, and its pass over the baseband -- a third read of the block -- is saved.  Nothing
  // of it is committed before the squelch sequencer below has decided (the sums and phase_memory are locals, the baseband goes to the
  // channel's scratch block).
  double fmsnr = snr;
  const bool need_var = active && !(snr_squelch || (st.squelch_state <= 0 && snr < sq_close));
  const unsigned long long var_rows = __ballot(need_var);
  if (var_rows != 0ull) {                                  // wave-uniform
    double avg = 0.0, var = 0.0;
    for (int pass = 0; pass < 2; pass++) {
      fetch_x(0, var_rows);
      for (int t0 = 0; t0 < N; t0 += LIN_TILE) {
        const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
        place_x();
        CHZ_WAVE_SYNC();
        if (t0 + LIN_TILE < N) fetch_x(t0 + LIN_TILE, var_rows);
        if (need_var) {
          int n = 0;
#if CHZ_FM_DISC_UNROLL > 1
          if (pass == 1)
            for (; n + CHZ_FM_DISC_UNROLL <= tn; n += CHZ_FM_DISC_UNROLL) {
              float2 v[CHZ_FM_DISC_UNROLL]; double ph[CHZ_FM_DISC_UNROLL];
#pragma unroll
              for (int u = 0; u < CHZ_FM_DISC_UNROLL; u++) v[u] = tile[lane * LD + n + u];
#pragma unroll
              for (int u = 0; u < CHZ_FM_DISC_UNROLL; u++) ph[u] = u ? disc_phase(v[u], (double)v[u - 1].x, (double)v[u - 1].y) : disc_phase(v[0], pr, pi);
#pragma unroll
              for (int u = 0; u < CHZ_FM_DISC_UNROLL; u++) {
                const double a = (double)demod_cabsf(v[u]);
                const double dlt = a - avg; var += dlt * dlt;
                tile[lane * LD + n + u].x = disc_commit(v[u], ph[u]);
              }
            }
#endif
          for (; n < tn; n++) {
            const float2 v = tile[lane * LD + n];
            const double a = (double)demod_cabsf(v);
            if (pass == 0) avg += a;
            else { const double dlt = a - avg; var += dlt * dlt; tile[lane * LD + n].x = discriminate(v); }
          }
        }
        CHZ_WAVE_SYNC();
        if (pass == 1) {
          for (int r0 = 0; r0 < 64; r0 += RPS) {
            const int rr = r0 + lane / LIN_TILE, n = lane % LIN_TILE;
            if (((var_rows >> rr) & 1ull) && n < tn) mixf[(size_t)(base + rr) * 2 * N + t0 + n] = tile[rr * LD + n].x;
          }
          CHZ_WAVE_SYNC();
        }
      }
      if (pass == 0) avg /= N;
    }
    if (need_var) {
      const double s2 = fm_snr_dev(avg * avg * (N - 1) / var);
      fmsnr = s2 > 0.0 ? s2 : 0.0;
    }
  }
  // ---- squelch sequencer (:149-173)
  const int smax = squelch_tail + 5;
  if (fmsnr >= sq_open) st.squelch_state = smax;
  else if (st.squelch_state > 0 && (fmsnr < sq_close || st.squelch_state < smax)) st.squelch_state--;
  DemodStatus r;
  r.pll_lock = 0; r.pll_snr = 0.0; r.pll_cphase = 0.0; r.pll_rotations = 0; r.tone_deviation = 0.0; r.tone_mute = 0;
  r.n0 = st.n0; r.snr = fmsnr; r.squelch_state = st.squelch_state; r.gain = 0.0;
  DemodExt* __restrict__ ext = p.ext + ch;                 // touched only with a tone squelch
  bool go = active;
  if (active && st.squelch_state <= 4) {
    if (st.squelch_state >= 1) { st.pm_re = 0.0; st.pm_im = 0.0; }
    r.frame = 1; r.mute = st.squelch_state == 0; r.output_power = 0.0; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
    if (tone) {
      if (st.squelch_state == 4) { ext->g_s0 = 0.0; ext->g_s1 = 0.0; }          // reset_goertzel
      if (st.squelch_state >= 1) ext->pl_sample_count = 0;
      r.tone_deviation = ext->tone_deviation; r.tone_mute = ext->tone_mute;
    }
    demod_publish(p, ch, r); p.state[ch] = st;
    go = false;
  }
  const unsigned long long go_rows = __ballot(go);
  if (go_rows == 0ull) return;                             // wave-uniform
  // ---- pass 3: the discriminator for the lanes that did not carry it through the variance pass (SNR squelch, or the squelch was shut
  // and has just been told to open by the cheap estimator)
  const bool late = go && !need_var;
  const unsigned long long late_rows = __ballot(late);
  if (late_rows != 0ull) {                                 // wave-uniform
    fetch_x(0, late_rows);
    for (int t0 = 0; t0 < N; t0 += LIN_TILE) {
      const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
      place_x();
      CHZ_WAVE_SYNC();
      if (t0 + LIN_TILE < N) fetch_x(t0 + LIN_TILE, late_rows);
      if (late) {
        int n = 0;
#if CHZ_FM_DISC_UNROLL > 1
        for (; n + CHZ_FM_DISC_UNROLL <= tn; n += CHZ_FM_DISC_UNROLL) {
          float2 v[CHZ_FM_DISC_UNROLL]; double ph[CHZ_FM_DISC_UNROLL];
#pragma unroll
          for (int u = 0; u < CHZ_FM_DISC_UNROLL; u++) v[u] = tile[lane * LD + n + u];
#pragma unroll
          for (int u = 0; u < CHZ_FM_DISC_UNROLL; u++) ph[u] = u ? disc_phase(v[u], (double)v[u - 1].x, (double)v[u - 1].y) : disc_phase(v[0], pr, pi);
#pragma unroll
          for (int u = 0; u < CHZ_FM_DISC_UNROLL; u++) tile[lane * LD + n + u].x = disc_commit(v[u], ph[u]);
        }
#endif
        for (; n < tn; n++) tile[lane * LD + n].x = discriminate(tile[lane * LD + n]);
      }
      CHZ_WAVE_SYNC();
      for (int r0 = 0; r0 < 64; r0 += RPS) {
        const int rr = r0 + lane / LIN_TILE, n = lane % LIN_TILE;
        if (((late_rows >> rr) & 1ull) && n < tn) mixf[(size_t)(base + rr) * 2 * N + t0 + n] = tile[rr * LD + n].x;
      }
      CHZ_WAVE_SYNC();
    }
  }
  if (go) { st.pm_re = pr; st.pm_im = pi; }                // phase_memory = the block's last sample
  st.pll_was_on = 0;                                       // :209
  if (go && st.squelch_state == smax) {                    // :232-256
    const double foff = psum * (samprate * 0.5 / N);
    double ppos = pmax, pneg = pmin;
    st.foffset += p.fm_alpha * (foff - st.foffset);
    ppos *= samprate * 0.5; pneg *= samprate * 0.5;
    ppos -= st.foffset; pneg -= st.foffset;
    st.pdeviation = ppos > -pneg ? ppos : -pneg;
  }
  const bool pm = deemph_rate != 0;
  const float dc = (float)(2 * st.foffset / samprate);     // :258-263 (applied with the de-emphasis)
  // ---- pass 4: PL / CTCSS tone squelch (:264-311) on the baseband after DC removal
  const unsigned long long tone_rows = __ballot(go && tone);
  if (tone_rows != 0ull) {
    FmTone g{0.0, 0.0, 0.0, 0.0, 0, 0};
    const int isamprate = (int)samprate;
    const int integrate = (int)rint(isamprate * 0.24);
    if (go && tone) g = FmTone{ext->g_s0, ext->g_s1, ext->old_pl_phase, ext->tone_deviation, ext->pl_sample_count, ext->tone_mute};
    fetch_b(0, tone_rows);
    for (int t0 = 0; t0 < N; t0 += LIN_TILE) {
      const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
      place_b();
      CHZ_WAVE_SYNC();
      if (t0 + LIN_TILE < N) fetch_b(t0 + LIN_TILE, tone_rows);
      if (go && tone)
        for (int n = 0; n < tn; n++) {
          float b = tilef[lane * LD + n];
          if (pm) b -= dc;
          fm_tone_sample(g, ct, (double)b, isamprate, integrate);
        }
      CHZ_WAVE_SYNC();
    }
    if (go && tone) {
      ext->g_s0 = g.s0; ext->g_s1 = g.s1; ext->old_pl_phase = g.old_phase; ext->pl_sample_count = g.count;
      ext->tone_mute = g.tmute; ext->tone_deviation = g.tdev;
      r.tone_deviation = g.tdev; r.tone_mute = g.tmute;
      if (g.tmute) {                                       // :305-309: muted before de-emphasis runs
        r.frame = 1; r.mute = 1; r.output_power = 0.0; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
        demod_publish(p, ch, r); p.state[ch] = st;
        go = false;
      }
    }
  }
  const unsigned long long out_rows = __ballot(go);
  if (out_rows == 0ull) return;                            // wave-uniform
  // ---- pass 5: de-emphasis (:312-320), gain (:325), PCM
  rows[lane] = LinRow{p.pcm + (size_t)ch * p.pcm_stride, enc, 1, go ? 1 : 0};
  const double gain = (2 * headroom * samprate) / bandwidth;
  double y = st.deemph_state, part = 0.0;
  // every row that sends is S16 and the PCM rows are 8-byte aligned (wave-uniform): the packed store below
  const bool fm_s16 = CHZ_LIN_PACKED_STORE && LIN_TILE == 16 && (p.pcm_stride & 7) == 0 && __ballot(go && !(enc == CHZ_PCM_S16BE_K || enc == CHZ_PCM_S16LE_K)) == 0ull;
  fetch_b(0, out_rows);
  for (int t0 = 0; t0 < N; t0 += LIN_TILE) {
    const int tn = N - t0 < LIN_TILE ? N - t0 : LIN_TILE;
    place_b();
    CHZ_WAVE_SYNC();
    if (t0 + LIN_TILE < N) fetch_b(t0 + LIN_TILE, out_rows);
    if (go) {
      auto out_step = [&](float b) -> float {              // de-emphasis as the recurrence it is, gain, power sum: in the reference's order
        if (pm) { b -= dc; y += deemph_rate * (deemph_gain * (double)b - y); b = (float)y; }
        const double sgn = gain * (double)b;
        part += sgn * sgn;
        return (float)sgn;
      };
      int n = 0;
#if CHZ_LIN_UNROLL > 1
      for (; n + CHZ_LIN_UNROLL <= tn; n += CHZ_LIN_UNROLL) {       // (see demod_lin_lanes: one LDS round trip per group instead of per sample)
        float v[CHZ_LIN_UNROLL];
#pragma unroll
        for (int u = 0; u < CHZ_LIN_UNROLL; u++) v[u] = tilef[lane * LD + n + u];
#pragma unroll
        for (int u = 0; u < CHZ_LIN_UNROLL; u++) v[u] = out_step(v[u]);
#pragma unroll
        for (int u = 0; u < CHZ_LIN_UNROLL; u++) tilef[lane * LD + n + u] = v[u];
      }
#endif
      for (; n < tn; n++) tilef[lane * LD + n] = out_step(tilef[lane * LD + n]);
    }
    CHZ_WAVE_SYNC();
    if (fm_s16 && tn == LIN_TILE) {                        // S16 rows leave as 8-byte words, four samples per lane (see demod_lin_lanes)
      for (int r0 = 0; r0 < 64; r0 += 16) {
        const int rr = r0 + (lane >> 2), n4 = (lane & 3) * 4;
        if ((out_rows >> rr) & 1ull) {
          const LinRow q = rows[rr];
          const bool be = q.enc == CHZ_PCM_S16BE_K;
          const unsigned long long w = (unsigned long long)demod_s16(tilef[rr * LD + n4], be) | ((unsigned long long)demod_s16(tilef[rr * LD + n4 + 1], be) << 16) |
                                       ((unsigned long long)demod_s16(tilef[rr * LD + n4 + 2], be) << 32) | ((unsigned long long)demod_s16(tilef[rr * LD + n4 + 3], be) << 48);
          *reinterpret_cast<unsigned long long*>(q.o + 2 * (t0 + n4)) = w;
        }
      }
    } else
    for (int r0 = 0; r0 < 64; r0 += RPS) {
      const int rr = r0 + lane / LIN_TILE, n = lane % LIN_TILE;
      if (((out_rows >> rr) & 1ull) && n < tn) { const LinRow q = rows[rr]; demod_put(q.o, q.enc, t0 + n, tilef[rr * LD + n]); }
    }
    CHZ_WAVE_SYNC();
  }
  if (!go) return;
  st.deemph_state = y;
  r.frame = 0; r.mute = 0; r.gain = gain; r.output_power = part / N; r.foffset = st.foffset; r.pdeviation = st.pdeviation;
  demod_publish(p, ch, r); p.state[ch] = st;
}

__global__ void __launch_bounds__(64) demod_linear_tail(DemodParams p) {
  HIP_DYNAMIC_SHARED(double, esh)                          // [N] per-sample energies (AGC slices), then [N] complex samples (PLL modes)
  const int lane = (int)threadIdx.x;
  const int lc = (int)blockIdx.x;
  if (lc >= p.nch) return;
  const int ch = p.ch0 + lc;
  const DemodChan c = p.chan[ch];
  if (!c.on) return;
  DemodState st = p.state[ch];
  float2* xs = reinterpret_cast<float2*>(esh + p.olen);
  if (c.kind == 1) {                                                           // wave-uniform
    if (p.fm_lanes != 0 && c.pll_enable == 0) return;                           // demod_fm_lanes has served this channel
    demod_fm_wave(p, c, st, ch, lane, esh, xs); return;
  }
  if (p.lin_lanes != 0 && (c.pll_enable == 0 || (p.mix != nullptr && p.lin_pll != 0))) return;   // demod_lin_lanes has served this channel
  const int N = p.olen;
  const int SEG = (N + 63) >> 6;
  const int n0 = lane * SEG;                               // first sample of this lane
  const int cnt = n0 >= N ? 0 : (N - n0 < SEG ? N - n0 : SEG);
  const float2* __restrict__ x = p.in + (size_t)ch * N;
  unsigned char* __restrict__ o = p.pcm + (size_t)ch * p.pcm_stride;
  const double bb_power = p.power[ch];
  // src/radio.c:1466-1473
  const double est = p.n0[ch];
  if (st.n0 != st.n0) st.n0 = est;
  else { const double diff = est - st.n0; st.n0 += p.power_alpha * diff; }
  // ---- coherent modes (src/linear.c:76-153): the PLL mixes the block down with its VCO, sample by sample, before anything else
  // looks at it.  The block goes to LDS, lane 0 walks it, the mixed samples replace it there.
  const bool pll = c.pll_enable != 0;                      // wave-uniform
  double pll_snr = 0.0, pll_cph = 0.0, pll_foff = 0.0; int pll_lock = 0, pll_rot = 0;
  if (pll && p.mix != nullptr && p.lin_pll != 0) {
    // pll_lanes has run this channel's loop already (one channel per lane): the mixed block and the loop's results are in memory
    const DemodExt* __restrict__ ext = p.ext + ch;
    const float2* __restrict__ m = p.mix + (size_t)ch * N;
    for (int i = 0; i < cnt; i++) xs[n0 + i] = m[n0 + i];
    pll_snr = ext->pll_snr; pll_cph = ext->pll_cphase; pll_foff = ext->foffset; pll_lock = ext->pll.lock; pll_rot = ext->pll_rotations;
    CHZ_WAVE_SYNC();
  } else if (pll) {
    DemodExt* __restrict__ ext = p.ext + ch;
    for (int i = 0; i < cnt; i++) xs[n0 + i] = x[n0 + i];
    CHZ_WAVE_SYNC();
    if (lane == 0) {
      PllState q = ext->pll;
      const int isamprate = (int)c.samprate;
      const int lock_limit = (int)rint(0.5 * isamprate);                      // DEFAULT_PLL_LOCKTIME (:6,:38-40)
      double bw = c.pll_loop_bw / isamprate;
      if (q.lock) bw *= 0.1;
      pll_set_params(q, bw, M_SQRT1_2);                                        // DEFAULT_PLL_DAMPING (:5)
      double signal = 0.0, noise = 0.0;
      pll_foff = ext->foffset;
      for (int n = 0; n < N; n++) {
        double sn, cs; pll_nco(q.vco_phase, sn, cs);
        const float2 v = xs[n];
        const double br = v.x, bi = v.y;
        const double sr = br * cs + bi * sn, si = bi * cs - br * sn;           // buffer[n] * conj(vco)
        xs[n] = make_float2((float)sr, (float)si);
        double phase;
        if (q.lock) {
          if (!c.pll_square) { const double mag = sqrt(sr * sr + si * si); phase = (mag > 0) ? si / mag : 0.0; }
          else phase = sr * si / (sr * sr - si * si);
        } else {
          if (!c.pll_square) phase = atan2(si, sr);
          else phase = 0.5 * atan2(sr * si + si * sr, sr * sr - si * si);      // carg(s*s)
        }
        phase /= (2 * M_PI);
        pll_foff = isamprate * pll_run(q, phase);
        signal += sr * sr; noise += si * si;
      }
      pll_cph = ldexp(2 * M_PI * (double)q.vco_phase, -32);
      pll_rot = q.wraps;
      if (noise != 0) { pll_snr = (signal / noise) - 1; if (pll_snr < 0) pll_snr = 0; }
      else pll_snr = __builtin_nan("");
      if (pll_snr < c.squelch_close) {
        q.lock_count -= N;
        if (q.lock_count <= -lock_limit) { q.lock_count = -lock_limit; q.lock = 0; }
      } else if (pll_snr > c.squelch_open) {
        q.lock_count += N;
        if (q.lock_count >= lock_limit) {
          q.lock_count = lock_limit;
          if (!q.lock) { q.lock = 1; pll_rot = 0; }
        }
      }
      pll_lock = q.lock;
      ext->pll = q; ext->pll_snr = pll_snr; ext->pll_cphase = pll_cph; ext->foffset = pll_foff; ext->pll_rotations = pll_rot;
    }
    CHZ_WAVE_SYNC();
    pll_snr = __shfl(pll_snr, 0);                          // the squelch decisions below are taken by every lane
  }
  // chan->shift (src/linear.c:168-172): this lane's phasor at its first sample from the closed form, then stepped in double
  const bool rot = c.osc_freq != 0.0;
  double c0 = 1.0, s0 = 0.0, c1 = 1.0, s1 = 0.0;
  if (rot) {
    const double g = (double)(p.job - c.osc_job0) * (double)N + (double)n0;
    double hi = g * c.osc_freq, lo = fma(g, c.osc_freq, -hi);
    hi -= rint(hi);
    sincospi(2.0 * (c.osc_phase0 + hi + lo), &s0, &c0);
    sincospi(2.0 * c.osc_freq, &s1, &c1);
  }
  // ---- AGC (src/linear.c:177-234)
  double gain_change = 1.0;
  if (c.agc) {
    const double bn = sqrt(c.bandwidth * st.n0);
    const double ampl = sqrt(bb_power);
    int sps = (int)rint(N * .002 / p.blocktime);
    sps = sps < 1 ? 1 : sps;
    {
      double cr = c0, sr = s0;
      for (int i = 0; i < cnt; i++) {
        float2 v = pll ? xs[n0 + i] : x[n0 + i];
        if (rot) {
          const double xr = v.x, xi = v.y;
          v = make_float2((float)(xr * cr - xi * sr), (float)(xr * sr + xi * cr));
          const double nc = cr * c1 - sr * s1; sr = cr * s1 + sr * c1; cr = nc;
        }
        float a = v.x * v.x, b = v.y * v.y;
        CHZ_ROUNDED_F32(a); CHZ_ROUNDED_F32(b);
        esh[n0 + i] = (double)(a + b);                                // cnrmf
      }
    }
    CHZ_WAVE_SYNC();
    // slices [k*sps, (k+1)*sps) with (k+1)*sps < N (src/linear.c:199: `while (n + samples_per_slice < N)`), summed in order
    double peak = 0.0;
    for (int k = lane; (k + 1) * sps < N; k += 64) {
      double energy = 0.0;
      for (int i = 0; i < sps; i++) energy += esh[k * sps + i];
      if (energy > peak) peak = energy;
    }
    double peak_level = sqrt(wave_max(peak) / sps);
    if (peak_level * st.gain > M_SQRT2 * c.headroom) {
      st.gain = M_SQRT2 * c.headroom / peak_level;
      gain_change = 1.0;
      st.hangcount = (int)rint(0.08 * c.samprate);
    } else if (ampl * st.gain > c.headroom) {
      const double newgain = c.headroom / ampl;
      if (newgain > 0) gain_change = pow(newgain / st.gain, 1.0 / N);
      st.hangcount = (int)rint(c.hangtime * c.samprate);
    } else if (bn * st.gain > c.threshold * c.headroom) {
      const double newgain = c.threshold * c.headroom / bn;
      if (newgain > 0) gain_change = pow(newgain / st.gain, 1.0 / N);
    } else if (st.hangcount > 0) {
      st.hangcount -= N;
    } else {
      gain_change = c.recov_ps;                                               // pow(recovery_rate, 1 / samprate)
    }
  }
  // ---- squelch sequencer (src/linear.c:313-352).  It is advanced AFTER the final pass in the reference's program order, but
  // the two do not interact, and knowing the frame type up front saves packing PCM nobody will send.
  double snr = __builtin_huge_val();
  if (c.snr_squelch) snr = (bb_power / (st.n0 * c.bandwidth)) - 1.0;
  else if (pll) snr = pll_snr;                                                 // :317-318
  const int smax = c.squelch_tail + 4;
  if (!(c.snr_squelch || pll) || snr >= c.squelch_open) st.squelch_state = smax;
  else if (st.squelch_state > 0 && snr < c.squelch_close) st.squelch_state--;
  const bool data = st.squelch_state >= 4;
  // ---- final pass (src/linear.c:236-311); the gain ramp and the carrier filter run whether or not the frame is sent
  const double k_env = M_SQRT1_2;
  double gain = st.gain;
  // this lane's first sample sees the gain after n0 steps of the ramp: gain_change^n0 by squaring over the bits of n0 (a dozen
  // multiplications; libm's pow() is ~180 instructions, and every lane of every channel paid it)
  if (gain_change != 1.0 && n0 > 0) {
    double f = 1.0, b = gain_change;
    for (unsigned e = (unsigned)(n0 < N ? n0 : N); e; e >>= 1) { if (e & 1u) f *= b; b *= b; }
    gain *= f;
  }
  // pass A (envelope modes with carrier removal only): this lane's samples as an affine map of the incoming filter state
  const bool dcfilt = c.env && c.dc_alpha != 0;
  double am_in = st.am_dc;
  if (dcfilt) {
    double A = 1.0, B = 0.0;                               // am_dc_out = A * am_dc_in + B over this lane's samples
    {
      double g = gain, cr = c0, sr = s0;
      for (int i = 0; i < cnt; i++) {
        float2 v = pll ? xs[n0 + i] : x[n0 + i];
        if (rot) {
          const double xr = v.x, xi = v.y;
          v = make_float2((float)(xr * cr - xi * sr), (float)(xr * sr + xi * cr));
          const double nc = cr * c1 - sr * s1; sr = cr * s1 + sr * c1; cr = nc;
        }
        const double s = g * k_env * (double)demod_cabsf(v);
        g *= gain_change;
        A *= (1.0 - c.dc_alpha); B = (1.0 - c.dc_alpha) * B + c.dc_alpha * s;
      }
    }
    // inclusive scan of the maps over the lanes (composition: later o earlier), then shift by one lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double Ap = __shfl_up(A, d), Bp = __shfl_up(B, d);
      if (lane >= d) { B = A * Bp + B; A = A * Ap; }
    }
    const double Ae = __shfl_up(A, 1), Be = __shfl_up(B, 1);
    am_in = lane == 0 ? st.am_dc : Ae * st.am_dc + Be;
    // the state after the block is the last lane's composition applied to the old state
    st.am_dc = __shfl(A, 63) * st.am_dc + __shfl(B, 63);
  }
  double part = 0.0;
  {
    double cr = c0, sr = s0, am_dc = am_in;
    const int enc = c.encoding;
    for (int i = 0; i < cnt; i++) {
      const int n = n0 + i;
      float2 v = pll ? xs[n] : x[n];
      if (rot) {
        const double xr = v.x, xi = v.y;
        v = make_float2((float)(xr * cr - xi * sr), (float)(xr * sr + xi * cr));
        const double nc = cr * c1 - sr * s1; sr = cr * s1 + sr * c1; cr = nc;
      }
      if (c.channels == 1) {
        double s;
        if (c.env) {
          s = gain * k_env * (double)demod_cabsf(v);
          gain *= gain_change;
          part += s * s;
          if (dcfilt) { am_dc += c.dc_alpha * (s - am_dc); s -= am_dc; }
        } else {
          s = gain * (double)v.x;
          gain *= gain_change;
          part += s * s;
        }
        if (data) demod_put(o, enc, n, (float)s);
      } else {
        double a, b;
        if (c.env) {
          const double k = gain * k_env;
          a = k * (double)v.x; b = k * (double)demod_cabsf(v);
          gain *= gain_change;
          part += a * a + b * b;
          if (dcfilt) { am_dc += c.dc_alpha * (b - am_dc); b -= am_dc; }
        } else {
          a = gain * (double)v.x; b = gain * (double)v.y;
          gain *= gain_change;
          part += a * a + b * b;
        }
        if (data) { demod_put(o, enc, 2 * n, (float)a); demod_put(o, enc, 2 * n + 1, (float)b); }
      }
    }
  }
  // the block's final gain is what the lane holding the last sample ends with
  const int last_lane = (N - 1) / SEG;
  st.gain = __shfl(gain, last_lane);
  double output_power = wave_sum(part) / N;
  if (c.channels == 1) output_power *= 2;
  DemodStatus r;
  r.gain = st.gain; r.n0 = st.n0; r.snr = snr; r.squelch_state = st.squelch_state;
  r.output_power = output_power; r.foffset = pll_foff; r.pdeviation = 0.0;
  r.pll_lock = pll_lock; r.pll_snr = pll_snr; r.pll_cphase = pll_cph; r.pll_rotations = pll_rot; r.tone_deviation = 0.0; r.tone_mute = 0;
  if (!data) {
    r.frame = 1; r.mute = st.squelch_state == 0;
    if (st.squelch_state == 3 || st.squelch_state == 0) r.output_power = 0;
  } else {
    if (c.snr_squelch || pll) {
      if (snr < c.squelch_close) st.squelch_open = 0;
      else if (!st.squelch_open && snr > c.squelch_open) { st.squelch_open = 1; st.am_dc = 0; }
    } else st.squelch_open = 1;
    r.frame = 0;
    r.mute = (output_power == 0 || !st.squelch_open || !c.tuned);
  }
  if (lane == 0) { demod_publish(p, ch, r); p.state[ch] = st; }
}

// ------------------------------------------------------------------------------
// Small inline masters (radiod's filter2, src/radio.c:1572-1594: a private COMPLEX master of
// N = round2(2 * blocksize) points with ONE same-size COMPLEX slave, shift 0, optionally ISB, run inline by
// each channel thread after the first filter).  Thousands of them exist, each far too small for a launch of
// its own: ONE launch serves every instance that is due, one workgroup per instance, everything in LDS:
//   window (N samples, history + new, as the host ring holds them contiguously; src/filter.c:626-636)
//   -> forward N-point transform (src/filter.c:573-582)  -> gather with the slave's descriptor x response,
//   ISB unpacking, Nyquist bin zeroed (src/filter.c:728-793,895-911)  -> backward transform (:914), keep the
//   last olen samples (:357).
// Stateless: the overlap lives in the caller's ring, so an instance can be re-run or served by several slaves.
// The transform is a Stockham autosort over radix-{2,3,4,5,7,11,13} stages (natural order in, natural order out, two
// LDS buffers), twiddles from a float64-rounded table W_N^k.  N <= 8192, no prime factor above 13.
// ------------------------------------------------------------------------------
#ifndef CHZ_MINI_MAX_STAGES
#define CHZ_MINI_MAX_STAGES 14
#endif
struct MiniReq { ChanDesc d; int isb; int pad; };      // per request: gather descriptor (d.row = response row), ISB flag
struct MiniParams {
  const float2* in;       // [nreq][N] windows
  float2* out;            // [nreq][olen]
  const MiniReq* req;     // [nreq]
  const float2* resp;     // [rows][N]
  const float2* tw;       // [N]  e^{-2 pi i k / N}
  int N, olen, nstages;
  int radix[CHZ_MINI_MAX_STAGES];
};

template <int R, int SIGN>
__device__ __forceinline__ void mini_stage(const float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw,
                                           int N, int Ns, int tid, int nthr) {
  const int M = N / R;                 // butterflies in this stage
  const int tws = N / (Ns * R);        // table stride of W_(Ns*R)
  for (int j = tid; j < M; j += nthr) {
    const int k = j % Ns;
    float2 v[R];
    static_for<R>([&](auto t) {
      constexpr int T = decltype(t)::value;
      float2 a = x[j + T * M];
      if constexpr (T > 0) {
        float2 w = tw[(T * k * tws) % N];          // < N already (T*k < Ns*R), the modulo only guards the table
        if (SIGN > 0) w.y = -w.y;
        a = cmul(a, w);
      }
      v[T] = a;
    });
    reg_dft<R, SIGN>(v);
    const int j0 = (j / Ns) * Ns * R + k;
    static_for<R>([&](auto t) { constexpr int T = decltype(t)::value; y[j0 + T * Ns] = v[T]; });
  }
}
template <int SIGN>
__device__ __forceinline__ float2* mini_fft(float2* a, float2* b, const MiniParams& p, int tid, int nthr) {
  int Ns = 1;
  for (int s = 0; s < p.nstages; s++) {
    const int r = p.radix[s];
    if (r == 2) mini_stage<2, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    else if (r == 3) mini_stage<3, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    else if (r == 4) mini_stage<4, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    else if (r == 5) mini_stage<5, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    else if (r == 7) mini_stage<7, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    else if (r == 11) mini_stage<11, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    else mini_stage<13, SIGN>(a, b, p.tw, p.N, Ns, tid, nthr);
    Ns *= r;
    __syncthreads();
    float2* t = a; a = b; b = t;
  }
  return a;        // where the result is
}

__global__ void __launch_bounds__(256) mini_ovs(MiniParams p) {
  HIP_DYNAMIC_SHARED(float2, lds)
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int N = p.N;
  float2* A = lds;
  float2* B = lds + N;
  const MiniReq rq = p.req[blockIdx.x];
  const float2* __restrict__ win = p.in + (size_t)blockIdx.x * N;
  for (int i = tid; i < N; i += nthr) A[i] = win[i];
  __syncthreads();
  float2* X = mini_fft<-1>(A, B, p, tid, nthr);            // master spectrum, natural order
  float2* Y = (X == A) ? B : A;
  // gather x response into the slave's bins (FFT order), exactly as chan_ifft does for a COMPLEX master
  const float2* __restrict__ H = p.resp + (size_t)rq.d.row * N;
  const int P = N;
  for (int i = tid; i < P; i += nthr) {
    int t = i - (P + 1) / 2; if (t < 0) t += P;            // rank from the most negative bin
    const int u = t - rq.d.t0;
    const bool ok = (u >= 0) && (u < rq.d.cnt);
    int src = rq.d.src0 + rq.d.dir * u;
    if (rq.d.wrap && src >= rq.d.wrap) src -= rq.d.wrap;
    float2 v = make_float2(0.f, 0.f);
    if (ok) { v = X[src]; if (rq.d.conj) v.y = -v.y; v = cmul(v, H[i]); }
    Y[i] = v;
  }
  __syncthreads();
  if (rq.isb) {                                            // src/filter.c:895-909
    for (int q = tid; q <= P / 2; q += nthr) {
      if (q == 0) { Y[0] = make_float2(0.f, 0.f); continue; }
      if (2 * q >= P) continue;
      const float2 pos = Y[q], neg = Y[P - q];
      Y[q] = make_float2(pos.x + neg.x, pos.y - neg.y);          // pos + conj(neg)
      Y[P - q] = make_float2(neg.x - pos.x, neg.y + pos.y);      // neg - conj(pos)
    }
    __syncthreads();
  }
  if (tid == 0) Y[(P + 1) / 2] = make_float2(0.f, 0.f);     // :911 comes after the unpack
  __syncthreads();
  float2* Z = mini_fft<+1>(Y, X, p, tid, nthr);
  float2* __restrict__ o = p.out + (size_t)blockIdx.x * p.olen;
  const int drop = P - p.olen;
  for (int i = tid; i < p.olen; i += nthr) o[i] = Z[drop + i];
}

// ------------------------------------------------------------------------------
// K3+K4 for ANY P without a prime factor above 13 -- the sizes the register-tiled chan_ifft / chan_c2r menu does not hold
// (wfm's 384 kHz channel on a 20 ms block with overlap 5 is P = 9600, src/wfm.c:37-39; odd sample rates).  One workgroup
// per channel: the gather x response of src/filter.c:728-911 lands in LDS in FFT order (COMPLEX or REAL output, ISB
// unpacking, Nyquist bin zeroed -- the same rules as chan_ifft / chan_c2r, bin by bin), the backward transform is the
// Stockham stage loop of mini_ovs, the last olen samples leave with the optional downconvert() epilogue.
// ------------------------------------------------------------------------------
// BIG: P beyond what two buffers in LDS hold (10240 < P <= CHZ_ANY_MAX_P = 2^20: 768 kHz channels and wider): the two buffers live in a
// per-workgroup piece of global scratch instead, which the L2 keeps; everything else is the same code.  (Within a workgroup
// a barrier orders global memory as it orders LDS: the wavefronts of a workgroup share their CU's vector cache.)
// Bluestein (P with a prime factor above 13): m describes the M-point transform (M = 2^k >= 2P - 1); m.tw holds W_M [M], then the chirp
// w_n = e^{+i pi n^2 / P} [P], then F(b) [M] with b_m = conj(w_m) wrapped around M.  With a = Y w (the gathered, filtered bins
// times the chirp), the P-point backward transform is  X_n = w_n * (a (*) b)_n : two M-point transforms with the one set of
// backward stages (the forward one as conj B conj), a pointwise product, a second chirp.
struct AnyParams { ChanParams c; MiniParams m; int real_out; float2* scratch; int P; };

template <bool BIG>
__global__ void __launch_bounds__(1024) chan_any(AnyParams q) {
  HIP_DYNAMIC_SHARED(float2, lds)
  const ChanParams& p = q.c;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int P = q.P;
  const int LB = q.m.N;                                     // points per buffer: P, or Bluestein's M
  if ((int)blockIdx.x >= p.nch) return;
  const int ch = p.ch0 + (int)blockIdx.x;
  float2* A = BIG ? q.scratch + (size_t)blockIdx.x * 2 * (size_t)LB : lds;
  float2* B = A + LB;
  const ChanDesc d = p.desc[ch];
  const float2* __restrict__ H = p.resp + (long)d.row * P;
  const float2* __restrict__ X = p.spec;
  if (!q.real_out) {
    for (int i = tid; i < P; i += nthr) {
      int t = i - (P + 1) / 2; if (t < 0) t += P;          // rank from the most negative bin
      const int u = t - d.t0;
      const bool ok = (u >= 0) && (u < d.cnt) && (i != (P + 1) / 2);
      int src = d.src0 + d.dir * u;
      if (d.wrap && src >= d.wrap) src -= d.wrap;
      float2 v = make_float2(0.f, 0.f);
      if (ok) { v = X[spec_addr(p.lay, src)]; if (d.conj) v.y = -v.y; v = cmul(v, H[i]); }
      A[i] = v;
    }
    __syncthreads();
    if (p.isb != nullptr && p.isb[ch] != 0) {              // src/filter.c:895-909 (workgroup-uniform)
      for (int k = tid; k <= P / 2; k += nthr) {
        if (k == 0) { A[0] = make_float2(0.f, 0.f); continue; }
        if (2 * k >= P) continue;
        const float2 pos = A[k], neg = A[P - k];
        A[k] = make_float2(pos.x + neg.x, pos.y - neg.y);          // pos + conj(neg)
        A[P - k] = make_float2(neg.x - pos.x, neg.y + pos.y);      // neg - conj(pos)
      }
      __syncthreads();
      if (tid == 0) A[(P + 1) / 2] = make_float2(0.f, 0.f);        // :911 comes after the unpack
      __syncthreads();
    }
  } else {
    const int SB = P / 2 + 1, shift = d.shift;
    for (int i = tid; i < P; i += nthr) {
      const int k = i <= P / 2 ? i : P - i;                 // the slave bin this entry of the Hermitian extension comes from
      const int mi = k + shift;
      bool ok; int a, b = 0;
      if (p.m_real) { ok = mi >= 0 && mi < p.m_bins; a = ok ? mi : 0; }                         // :808
      else {
        ok = mi >= -(p.m_bins / 2) && mi < p.m_bins / 2;                                        // :798
        a = mi % p.m_bins; if (a < 0) a += p.m_bins;
        b = (p.m_bins - mi) % p.m_bins; if (b < 0) b += p.m_bins;
      }
      if (k == (SB + 1) / 2) ok = false;                                                        // :911
      float2 x = make_float2(0.f, 0.f);
      if (ok) {
        x = X[spec_addr(p.lay, a)];
        if (!p.m_real) { const float2 w = X[spec_addr(p.lay, b)]; x = make_float2(x.x + w.x, x.y - w.y); }   // X[a] + conj(X[b]), :800
        x = cmul(x, H[k]);
        if (k == 0 || 2 * k == P) x.y = 0.f;                // c2r ignores these imaginary parts
        if (i > P / 2) x.y = -x.y;
      }
      A[i] = x;
    }
    __syncthreads();
  }
  float2* Y; float2* W;
  const int drop = P - p.olen;
  if (LB == P) {
    Y = mini_fft<+1>(A, B, q.m, tid, nthr);
    W = (Y == A) ? B : A;                                   // free again: reduction scratch
  } else {                                                  // workgroup-uniform: Bluestein
    const float2* __restrict__ chirp = q.m.tw + LB;
    const float2* __restrict__ fb = chirp + P;
    for (int i = tid; i < LB; i += nthr) {                  // conj(a), zero-padded to M
      float2 v = make_float2(0.f, 0.f);
      if (i < P) { v = cmul(A[i], chirp[i]); v.y = -v.y; }
      A[i] = v;
    }
    __syncthreads();
    float2* X1 = mini_fft<+1>(A, B, q.m, tid, nthr);        // B(conj a) = conj F(a)
    float2* O1 = (X1 == A) ? B : A;
    for (int j = tid; j < LB; j += nthr) { float2 v = X1[j]; v.y = -v.y; X1[j] = cmul(v, fb[j]); }   // F(a) F(b)
    __syncthreads();
    float2* X2 = mini_fft<+1>(X1, O1, q.m, tid, nthr);      // M (a (*) b)
    const float inv = 1.0f / (float)LB;
    for (int n = drop + tid; n < P; n += nthr) { const float2 v = cmul(X2[n], chirp[n]); X2[n] = make_float2(v.x * inv, v.y * inv); }   // in place
    __syncthreads();
    Y = X2; W = (X2 == A) ? B : A;
  }
  if (q.real_out) {
    float* __restrict__ o = reinterpret_cast<float*>(p.out) + (long)ch * p.olen;
    for (int n = tid; n < p.olen; n += nthr) o[n] = Y[drop + n].x;
    return;
  }
  float2* __restrict__ o = p.out + (long)ch * p.olen;
  const bool fine = p.fine != nullptr && p.fine[ch].on;
  double part = 0.0;
  if (fine) {
    const FineDesc f = p.fine[ch];
    const double kb = (double)(p.job - f.job0);
    const unsigned r = (unsigned)(((unsigned long long)((p.job - f.job0) % (unsigned)f.V + 1u) * (unsigned)f.adj_num) % (unsigned)f.V);
    const double base = f.phase0 + (double)r / (double)f.V;
    for (int n = tid; n < p.olen; n += nthr) {
      const double g = kb * (double)p.olen + (double)n;
      double hi = g * f.freq, lo = fma(g, f.freq, -hi);
      hi -= rint(hi);
      double qd = 0.0;
      if (f.rate != 0.0) { qd = 0.5 * f.rate * g * (g + 1.0); qd -= rint(qd); }
      double sn, cs;
      sincospi(2.0 * (base + hi + lo + qd), &sn, &cs);
      const float2 v = Y[drop + n];
      const double xr = v.x, xi = v.y;
      const float2 w = make_float2((float)(xr * cs - xi * sn), (float)(xr * sn + xi * cs));
      o[n] = w;
      part += (double)(w.x * w.x + w.y * w.y);
    }
  } else {
    for (int n = tid; n < p.olen; n += nthr) {
      const float2 v = Y[drop + n];
      o[n] = v;
      part += (double)(v.x * v.x + v.y * v.y);
    }
  }
  if (p.power != nullptr) {                                 // workgroup-uniform
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) part += __shfl_xor(part, dd);
    double* red = reinterpret_cast<double*>(W);
    __syncthreads();                                        // every lane is done with Y's partner buffer
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < (nthr + 63) / 64; w++) tot += red[w];
      p.power[ch] = tot / (double)p.olen;
    }
  }
}

}  // namespace chz
