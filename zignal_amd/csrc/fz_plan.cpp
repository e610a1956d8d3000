// Which kernel variant runs: the library's static choice (resolve_variant / finalize_variant).  Measured plans, their persistence per
// board and the measurement itself: fz_tune.cpp.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "fz_runtime.hpp"

namespace fz {

// ---- variant selection ---------------------------------------------------------------------------------
constexpr uint64_t kMaxLdsBytes = 160 * 1024;
// Output rows that start off this grid are stored with the merging policy (FZ_VF_ST_MERGE).  Measured, 6-biquad cascade x 4096 samples, nt | sc1
// against nt alone: rows on the 4-byte grid (1 000 001 streams) 6.94 / 5.90 ms, 16-byte (1 000 004) 6.21 / 5.50, 32-byte (1 000 008) 5.86 / 5.52,
// 64-byte (1 000 016) 5.39 / 5.43, 128-byte 5.42 / 5.43, 1 048 576 streams 5.48 / 5.52 (profiles/r04/rows_off_the_grid_store_policy.txt)
constexpr uint64_t kStoreGridBytes = 64;
constexpr uint32_t kLockstepMinRows = 1024;

// tile_streams: the frame layout the variant is for -- 0 (or >= n_streams) plain time-major rows, else stream tiles (stream-major
// frames are named by FZ_VF_STREAM_MAJOR in the variant's flags).  allow_lockstep: see the time-major rule below.
// (stream-major frames: nothing asked for besides the layout)
static bool uv_has_shape(const fz_variant* uv)
{
   return uv && (uv->streams_per_lane || uv->unroll || uv->block_threads || (uv->flags & ~(uint32_t)FZ_VF_STREAM_MAJOR));
}

// ---- plain time-major frames of many streams: the launch geometry ------------------------------------------------------------
// The row walk in lockstep (FZ_VF_LOCKSTEP + FZ_VF_GRID_SYNC: one workgroup per CU that meets at a barrier after every chunk, the
// workgroups of an XCD in step through arrival counters) has three knobs: streams per lane P, lanes per workgroup, and -- with
// more work than one workgroup per CU -- LAPS, each a launch over a contiguous stream range.  They follow from the chip (CUs,
// 1024 lanes per workgroup, whole waves per SIMD: lanes in multiples of 256 -- 960 lanes measured 0.50 of peak against 0.74 for
// 1024) and from what each (P, lanes) pair streams, measured on the 6-biquad cascade x 4096 samples with every workgroup filled
// (profiles/r04/sweep_time_major_geometry.txt; fraction of 8 TB/s):
//        lanes      256     512     768    1024
//        P = 4     0.64    0.76    0.76    0.76        (U = 4 / 2 / 1 / 1: the rows ahead per CU stay >= 16 KiB)
//        P = 2     0.61    0.74    0.72    0.78
//        P = 1      --      --      --     0.65 (the cascade is VALU-bound there: 54 scalar operations per sample; light graphs 0.72)
// A block of several laps loses about 5 % (2 097 152 streams: 0.75 in four laps of P = 2, 0.70 in two of P = 4; 1 572 864: 0.76 either
// way).  The choice is the candidate with the best  table value x fill (streams / (laps x CUs x lanes x P)) x lap penalty.
// A stream count just above whole laps (1 048 577: one stream more than 256 workgroups of 1024 lanes x 4 hold) would pay a whole extra
// lap for its last few streams; such a block runs as whole laps plus a REMAINDER launch (at most 65 536 streams: the few-stream
// kernels, ~0.35 ms per 4096 samples whatever the count -- 6 % of a lap of a million streams).  A count that is not a multiple of P
// but fits the workgroups is no obstacle either: FZ_VF_RAGGED.
TmGeometry time_major_geometry(uint64_t n_streams, uint32_t max_p, bool heavy_ops, bool ragged_ok, uint32_t only_p)
{
   static const double eff[3][4] = {{0.55, 0.60, 0.62, 0.72}, {0.61, 0.74, 0.72, 0.78}, {0.64, 0.76, 0.76, 0.76}};   // [P = 1, 2, 4][lanes / 256 - 1]
   constexpr double kRemainderMs = 0.35, kBytesPerMs = 8.0e9, kBytesPerStream = 32880.0;   // (per 4096-sample block: the unit of the comparison)
   constexpr uint64_t kRemainderMax = 65536;
   const uint64_t cus = chip_cus();
   TmGeometry best;
   auto consider = [&](uint32_t P, uint64_t lanes, uint64_t laps, uint64_t main_streams, double ms) {
      const double sc = (double)n_streams * kBytesPerStream / (ms * kBytesPerMs);
      if (sc <= best.score) return;
      best.P = P;
      best.lanes = (uint32_t)lanes;
      best.laps = (uint32_t)laps;
      best.main_streams = main_streams;
      best.score = sc;
      // rows ahead per CU: a CU's piece of a row is lanes x P x 4 bytes -- 12 KiB and more: one row per buffer, three buffers (two
      // rows ahead); 6-8 KiB: chunks of two rows; 4 KiB: of four
      const uint64_t piece = lanes * P * 4u;
      best.U = piece >= 12288 ? 1u : piece >= 6144 ? 2u : 4u;
   };
   for (uint32_t P = 4, pi = 2; P >= 1; P /= 2, --pi) {
      if (P > max_p || (only_p && P != only_p)) continue;
      auto ms_of = [&](uint64_t lanes, uint64_t laps) {
         double e = eff[pi][lanes / 256 - 1] * (laps > 1 ? 0.95 : 1.0);
         if (P == 1 && heavy_ops) e *= 0.9;
         return (double)(laps * cus * lanes * P) * kBytesPerStream / (e * kBytesPerMs);
      };
      if (n_streams % P == 0 || ragged_ok) {                 // all streams in laps of equal size
         const uint64_t groups = (n_streams + P - 1) / P;
         const uint64_t laps = std::max<uint64_t>(1, (groups + cus * 1024 - 1) / (cus * 1024));
         const uint64_t lanes = std::min<uint64_t>(std::max<uint64_t>(((groups + laps * cus - 1) / (laps * cus) + 255) / 256 * 256, 256), 1024);
         consider(P, lanes, laps, n_streams, ms_of(lanes, laps));
      }
      for (uint64_t lanes = 1024; lanes >= 256; lanes -= 256) {   // whole laps of full workgroups + a remainder launch
         const uint64_t per_lap = cus * lanes * P, laps = n_streams / per_lap;
         if (laps == 0) continue;
         const uint64_t rem = n_streams - laps * per_lap;
         if (rem == 0 || rem > kRemainderMax) continue;
         consider(P, lanes, laps, laps * per_lap, ms_of(lanes, laps) + kRemainderMs);
      }
   }
   return best;
}

// ---- the library's choice, rule by rule ----------------------------------------------------------------------------------------
// resolve_variant = checks of what the caller asked for, then ONE of three resolvers by kernel body: wave split, stream-major, frames.
// Every threshold below is a measurement; where it came from is in profiles/NOTES.md ("Planner rules"), not here.
struct Request {
   uint32_t P, U, B;                                       // streams per lane, unroll, block as asked for (0 = the library's choice)
   const fz_variant* uv;
};

static void check_request(const Graph& g, const Request& rq, const Variant& v)
{
   if (rq.P != 0 && rq.P != 1 && rq.P != 2 && rq.P != 4) fail(FZ_E_INVALID, "streams_per_lane must be 0, 1, 2 or 4");
   if (v.flags & FZ_VF_RESERVED) fail(FZ_E_INVALID, "variant flags: reserved bits set (include/flowz_hip.h lists the FZ_VF_* a caller can set)");
   if (g.typed && (v.flags & FZ_VF_OUT_F64))
      fail(FZ_E_INVALID, "FZ_VF_OUT_F64 does not apply to fz_compile_typed programs: their frames carry every wire in its own type");
   // (64 rows per chunk: frame kernels whose occupancy is capped by LDS rings -- one wave per SIMD keeps all the rows in flight itself)
   const bool long_unroll_ok = ((v.flags & FZ_VF_SM_LONG) && (rq.U == 64 || rq.U == 128)) ||
                               (rq.U == 64 && g.n_lds_slots && !(v.flags & FZ_VF_STREAM_MAJOR) && !ws_parts(v.flags));
   if (rq.U > 32 && !long_unroll_ok) fail(FZ_E_INVALID, "unroll must be <= 32");
   if (rq.B != 0 && (rq.B % 64 != 0 || rq.B > 1024)) fail(FZ_E_INVALID, "block_threads must be a multiple of 64, <= 1024");
   if ((v.flags & FZ_VF_LOCKSTEP) && (ws_parts(v.flags) || (v.flags & FZ_VF_STREAM_MAJOR)))
      fail(FZ_E_INVALID, "FZ_VF_LOCKSTEP applies to the frame kernel (time-major / tiled frames, lane-packed or stage-packed; no wave split)");
   if ((v.flags & FZ_VF_GRID_SYNC) && !(v.flags & FZ_VF_LOCKSTEP)) fail(FZ_E_INVALID, "FZ_VF_GRID_SYNC goes with FZ_VF_LOCKSTEP");
   if ((v.flags & FZ_VF_IO_WAVE2) && !(v.flags & FZ_VF_IO_WAVE)) fail(FZ_E_INVALID, "FZ_VF_IO_WAVE2 goes with FZ_VF_IO_WAVE");
}

// W compute waves per 64 streams, each evaluating one part of the serial graph (fz_split.cpp: find_wave_roles), and with FZ_VF_IO_WAVE
// one or two more waves for the frame I/O.  The waves of a workgroup go to consecutive SIMDs of a CU: as many tuples per workgroup as
// put one compute wave on every SIMD.
static Variant resolve_wave_split(const Graph& g, const Request& rq, Variant v)
{
   const uint32_t W = ws_parts(v.flags), waves = ws_waves(v.flags);
   if (!g.wave_roles(W))
      fail(FZ_E_UNSUPPORTED, W == 1 ? "FZ_VF_IO_WAVE: the graph is not stage-packable (1 in, 1 out, register delay lines)"
                                    : "wave split: the graph is not that many groups of isomorphic segments in series (1 in, 1 out, register delay lines)");
   if (rq.P > 1) fail(FZ_E_INVALID, "wave split needs streams_per_lane == 1");
   if (rq.B && (rq.B % 64 || rq.B * waves > 1024)) fail(FZ_E_INVALID, "wave split: block_threads counts the streams of a workgroup: a multiple of 64, at most 1024 / waves per tuple");
   if (rq.U && rq.U != 8 && rq.U != 16 && rq.U != 32) fail(FZ_E_INVALID, "wave split: unroll must be 8, 16 or 32");
   if (v.flags & (FZ_VF_STREAM_MAJOR | FZ_VF_OUT_F64 | FZ_VF_PREFETCH3))
      fail(FZ_E_UNSUPPORTED, "wave split: time-major / tiled float32 frames, double buffering only");
   v.P = 1;
   v.U = rq.U ? rq.U : (W == 1 ? 16 : 32);                  // rounds of 32 steps (one barrier each); the lone compute wave's rings fill the LDS at 32
   v.block = rq.B ? rq.B : (ws_io(v.flags) ? (W == 1 ? 256 : W == 2 ? 128 : 64) : (W == 2 ? 128 : 64));
   // the rings of a workgroup must fit the CU's LDS: tuples x hand-offs x ring x 1 KiB (every ring of a tuple is sized for the widest
   // hand-off: U groups behind a part that lags by more than 4 samples, else U / 2)
   uint32_t kmax = 0;
   for (const Graph& r : *g.wave_roles(W)) kmax = std::max(kmax, r.split.atoms());
   const uint32_t ring = (kmax - 1 > 4 ? v.U : v.U / 2), nring = W - 1 + 2 * ws_io(v.flags);
   while ((uint64_t)(v.block / 64) * nring * ring * 1024 > kMaxLdsBytes && !rq.B && v.block > 64) v.block /= 2;
   if ((uint64_t)(v.block / 64) * nring * ring * 1024 > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "wave split: the hand-off rings do not fit the LDS with this unroll and block size");
   v.flags &= ~(uint32_t)(FZ_VF_STAGE_PACK | FZ_VF_NO_STAGE_PACK);   // (each part is stage-packed by itself)
   return v;
}

// Plain time-major frames of many streams: one workgroup per CU in lockstep, the CUs of an XCD in step.  From one wave per
// SIMD and CU of work on, blocks of >= 1024 rows, narrow frames (four / two / one stream per lane as the registers allow) or 3-8 wire
// frames (one stream per lane), register delay lines only, nothing asked for by the caller.
static bool lockstep_default(const Graph& g, Variant& v, uint64_t n_streams, uint32_t n_samples, uint32_t allow_lockstep)
{
   const bool wide = (g.n_in > 2 || g.n_out > 2) && g.n_in <= 8 && g.n_out <= 8 && !g.typed;
   // (delay lines in HBM rings walk along: their reads are prefetched a chunk ahead like frame rows -- chunks of at least two rows, no third
   //  buffer, whole lanes only.  The 300-sample comb at 1 M streams: 10.85 ms against 11.2 ms free-running side by side; over the bench lines of
   //  a day 0.73 ... 0.80 of peak against 0.65 ... 0.77 free-running)
   const bool far = !g.far_lines.empty();
   // LDS rings walk along too (round 6) -- at the geometry their rings allow: one stream per lane, 256 lanes (one wave per SIMD; the comb's rings
   // take 104 KiB of the CU's 160), 16-row chunks, the resident workgroups as one lap of many.  On six fresh allocations the 40 / 23-sample
   // combs at 1 M streams run 0.681-0.717 of peak against 0.651-0.702 free-running -- ahead on every one of them, +1 ... +5 %
   // (profiles/r06/placement.txt; 32-row chunks in step 0.638-0.695, 20 / 24-row 0.67-0.70, 12-row 0.63, 8-row 0.51-0.54, 128 lanes 0.45; a THIRD
   // chunk buffer wins 15 of 18 paired comparisons on two boards and loses 24 of 24 on two others, +- 1-2.5 % either way: not taken)
   if (g.n_lds_slots && n_samples >= kLockstepMinRows && n_streams >= (uint64_t)chip_cus() * 1024u && g.n_in <= 2 && g.n_out <= 2 && !g.typed && !far) {
      Variant w = v;
      w.P = 1;
      w.U = 16;
      w.block = 256;
      w.flags |= FZ_VF_LOCKSTEP | FZ_VF_GRID_SYNC;
      if ((uint64_t)ring_plan(g, w).slots * w.block * 4u <= kMaxLdsBytes) {
         v = w;
         return true;
      }
   }
   if (!(n_samples >= kLockstepMinRows && n_streams >= (uint64_t)chip_cus() * 1024u && ((g.n_in <= 2 && g.n_out <= 2) || wide) &&
         g.n_lds_slots == 0 && !((g.typed || far) && (n_streams % 4)) && !(far && (wide || g.far_min_read < 4))))
      return false;
   const uint32_t cap = wide ? 1u : allow_lockstep >= 3 ? 4u : allow_lockstep;
   const TmGeometry geo = time_major_geometry(n_streams, cap, g.n_ops > 30, !g.typed && !far);
   v.P = geo.P;
   // wide frames: one row per buffer in one lap; in several laps chunks of THREE rows (the 4-wire sum at 1 M streams, rows 16 MiB apart:
   // 14.09-14.17 ms against 14.52-14.55 with two, 14.8 with four, 14.4-14.6 with five to seven; profiles/r05/lane_groups.txt)
   v.U = wide ? (geo.laps > 1 ? 3u : 1u) : geo.U;
   v.block = geo.lanes;
   if (far) v.U = std::min(std::max(v.U, 2u), std::max(2u, g.far_min_read / 2));
   v.flags |= FZ_VF_LOCKSTEP | FZ_VF_GRID_SYNC | (v.U == 1 ? (uint32_t)FZ_VF_PREFETCH3 : 0u);
   if (v.P == 1 && g.split.ok && n_samples >= 16u * (g.split.atoms() - 1)) {   // one stream per lane and a series of isomorphic segments: stage-packed
      v.flags |= FZ_VF_STAGE_PACK;
      v.flags &= ~(uint32_t)FZ_VF_PREFETCH3;
      v.U = std::max(v.U, 4u);
   }
   return true;
}

// Stream-major frames (fz_run_block_stream_major): the pair long-run body for deep 1-in/1-out graphs on many streams, the one-stream
// long-run body for other 1-in/1-out graphs, else short chunks; one-wave workgroups for the long-run bodies (a CU refills wave by wave).
static Variant resolve_stream_major(const Graph& g, const Request& rq, Variant v, uint64_t n_streams, uint32_t n_samples)
{
   const fz_variant* uv = rq.uv;
   if (rq.P > 2) fail(FZ_E_INVALID, "stream-major frames take one or two streams per lane");
   if (rq.U % 4) fail(FZ_E_INVALID, "stream-major frames need unroll % 4 == 0");
   if (!g.far_lines.empty()) fail(FZ_E_UNSUPPORTED, "stream-major frames: delay lines beyond 256 samples are not supported");
   if (v.flags & (FZ_VF_OUT_F64 | FZ_VF_PREFETCH3)) fail(FZ_E_UNSUPPORTED, "stream-major frames: float32 frames, double buffering only");
   v.P = rq.P ? rq.P : 1u;
   // the pair body from 2^19 streams on.  (Rounds 3-5 also took it from 2^17 on where its workgroups filled the chip's rounds; at 262 144 streams the
   //  one-stream body was level or ahead on every bench line since -- +1 ... +5 %, never behind: profiles/NOTES.md "Round 6")
   const bool enough = n_streams >= (1u << 19);
   if (!uv_has_shape(uv) && g.n_in == 1 && g.n_out == 1 && g.n_lds_slots == 0 && g.far_lines.empty() && g.n_param <= 32 && g.n_mod == 0 &&
       !g.typed && g.n_ops > 27 && g.n_state <= 20 && enough && n_streams % 2 == 0 && n_samples >= 256) {
      v.P = 2;
      v.flags |= FZ_VF_SM_LONG;
   }
   // stage packing (one stream per lane) carries over; automatic only for graphs deep enough to be VALU-bound with one stream per lane
   if (v.P != 1 || !g.split.ok) v.flags &= ~(uint32_t)FZ_VF_STAGE_PACK;
   else if (!(uv && (uv->flags & FZ_VF_NO_STAGE_PACK)) && n_samples >= 16u * (g.split.atoms() - 1) && g.n_ops > 27) v.flags |= FZ_VF_STAGE_PACK;
   const uint32_t nw = std::max<uint32_t>(std::max(g.n_in, g.n_out), 1);
   const bool long_ok = g.n_in == 1 && g.n_out == 1 && v.P == 1 && g.n_lds_slots == 0 && (!g.split.ok || g.split.atoms() <= 13);
   const bool pair_ok = g.n_in == 1 && g.n_out == 1 && v.P == 2 && g.n_lds_slots == 0 && !g.typed;
   const bool want_short = uv && (uv->flags & FZ_VF_SM_SHORT);
   v.flags &= ~(uint32_t)FZ_VF_SM_SHORT;
   if ((v.flags & FZ_VF_SM_LONG) && v.P == 2) {               // pair long-run body: halves of 64 samples, 512-byte out-runs
      if (!pair_ok) fail(FZ_E_UNSUPPORTED, "FZ_VF_SM_LONG with two streams per lane: needs a 1-in/1-out float graph (not fz_compile_typed), no delay lines beyond 8 samples");
      if (rq.U && rq.U != 64) fail(FZ_E_INVALID, "FZ_VF_SM_LONG with two streams per lane: unroll must be 64");
      v.U = 64;
      if (!rq.B) v.block = 64;
      auto lds_pair = [&](const Variant& w) { return (uint64_t)(w.block / 64) * 64 * (2 * w.U + 4) * 4; };
      while (lds_pair(v) > kMaxLdsBytes && !rq.B && v.block > 64) v.block /= 2;
      if (lds_pair(v) > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "FZ_VF_SM_LONG: the LDS patches do not fit this block size");
      return v;
   }
   if (v.flags & FZ_VF_SM_LONG) {
      if (!long_ok) fail(FZ_E_UNSUPPORTED, "FZ_VF_SM_LONG: needs a 1-in/1-out graph, one stream per lane, no delay lines beyond 8 samples");
      if (rq.U && rq.U != 64 && rq.U != 128) fail(FZ_E_INVALID, "FZ_VF_SM_LONG: unroll must be 64 or 128");
   } else if (long_ok && !want_short && !rq.U && n_samples >= 256) {
      v.flags |= FZ_VF_SM_LONG;
   }
   if (v.flags & FZ_VF_SM_LONG) {                            // one-stream long-run body: 512-byte runs from 2048 samples on, else 256-byte
      v.U = rq.U ? rq.U : (n_samples >= 2048 ? 128 : 64);
      if ((v.flags & FZ_VF_STAGE_PACK) && !g.split.ok) v.flags &= ~(uint32_t)FZ_VF_STAGE_PACK;
      auto lds_long = [&](const Variant& w) { return (uint64_t)(w.block / 64) * 64 * (w.U + 12) * 4; };
      if (!rq.B) v.block = 64;
      while (lds_long(v) > kMaxLdsBytes && !rq.B && v.block > 64) v.block /= 2;
      if (lds_long(v) > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "FZ_VF_SM_LONG: the LDS patches do not fit this block size");
      return v;
   }
   // short chunks: the deepest chunk whose patches fit the CU's LDS
   auto lds = [&](const Variant& w) { return (uint64_t)w.block * w.P * (w.U * nw + 4) * 4 + (uint64_t)g.n_lds_slots * w.block * 4 * w.P; };
   if (!rq.U) {
      v.U = 32;
      while (v.U > 4 && lds(v) > kMaxLdsBytes) v.U /= 2;
   }
   if ((v.flags & FZ_VF_STAGE_PACK) && v.U <= g.split.atoms() - 1) {
      if (uv && (uv->flags & FZ_VF_STAGE_PACK)) fail(FZ_E_INVALID, "stage-packed stream-major frames need unroll > number of segments - 1");
      v.flags &= ~(uint32_t)FZ_VF_STAGE_PACK;
   }
   while (lds(v) > kMaxLdsBytes && !rq.B && v.block > 64) v.block /= 2;
   if (lds(v) > kMaxLdsBytes) fail(FZ_E_UNSUPPORTED, "stream-major frames: the LDS patches do not fit (too many wires per frame)");
   // (two streams per lane with patches so large that a single wave fills the CU's LDS crawl: refuse)
   if (v.P == 2 && (uint64_t)64 * v.P * (v.U * nw + 4) * 4 > kMaxLdsBytes / 4)
      fail(FZ_E_UNSUPPORTED, "stream-major frames: two streams per lane leave one wave per CU with this many wires per frame and this "
                             "unroll; use one stream per lane or a shorter unroll");
   return v;
}

Variant resolve_variant(const Graph& g, const fz_variant* uv, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams, uint32_t allow_lockstep)
{
   if (tile_streams >= n_streams) tile_streams = 0;
   Variant v;
   const Request rq{uv ? uv->streams_per_lane : 0u, uv ? uv->unroll : 0u, uv ? uv->block_threads : 0u, uv};
   v.flags = uv ? uv->flags : 0;
   check_request(g, rq, v);
   if (ws_parts(v.flags)) return resolve_wave_split(g, rq, v);
   if (rq.P) {
      // (the lockstep frame kernel on plain time-major rows takes any count: FZ_VF_RAGGED)
      if (n_streams % rq.P && !((v.flags & FZ_VF_LOCKSTEP) && !tile_streams && !g.typed && g.far_lines.empty() && g.n_lds_slots == 0))
         fail(FZ_E_INVALID, "n_streams must be a multiple of streams_per_lane");
      v.P = rq.P;
   } else {
      // fill the chip first, then pack two streams per lane (narrow frames, from 2^18 even streams on); LDS rings: one stream per lane
      // (vectorised in time a 16-byte LDS access carries four time steps, and 256 lanes put a wave on every SIMD)
      v.P = (n_streams >= (1u << 18) && n_streams % 2 == 0 && g.n_in <= 2 && g.n_out <= 2) ? 2 : 1;
      if (g.n_lds_slots && !(v.flags & FZ_VF_STREAM_MAJOR)) v.P = 1;
   }
   const bool nothing_asked = !rq.P && !rq.U && !rq.B && !(v.flags & ~(uint32_t)FZ_VF_OUT_F64);
   if (allow_lockstep && nothing_asked && !tile_streams && lockstep_default(g, v, n_streams, n_samples, allow_lockstep)) return v;
   // deep graphs: register delay lines + prefetch buffers must stay inside the 512-entry register file
   uint32_t reg_state = 0;
   for (const Line& l : g.lines)
      if (!l.in_lds) reg_state += l.depth;
   if (!rq.P && v.P == 2 && reg_state > 36) v.P = 1;
   // rows per chunk: 16; 32 for wide frames of one stream per lane on an oversubscribed chip and for LDS rings; 8 for register-heavy lanes
   v.U = rq.U ? rq.U : 16;
   if (!rq.U && v.P == 1 && g.n_in >= 3 && n_streams >= (1u << 19) && !(v.flags & FZ_VF_STAGE_PACK)) v.U = 32;
   if (!rq.U && reg_state * v.P > 60) v.U = 8;
   else if (!rq.U && g.n_lds_slots && !(v.flags & FZ_VF_STAGE_PACK)) v.U = 32;
   if (!g.far_lines.empty()) {
      // far (HBM ring) reads are prefetched one chunk ahead: the chunk is at most half the youngest ring read
      const uint32_t cap = std::min(16u, std::max(1u, g.far_min_read ? g.far_min_read / 2 : 16u));
      if (rq.U > cap) fail(FZ_E_INVALID, "graphs with delays beyond LDS need unroll <= " + std::to_string(cap));
      if (v.flags & FZ_VF_PREFETCH3) fail(FZ_E_INVALID, "FZ_VF_PREFETCH3 is not available with delays beyond LDS");
      v.U = std::min(v.U, cap);
   }
   // few streams: the most parts whose waves still find a SIMD each (<= 16 384 streams: four or three, <= 32 768: two), with an I/O wave;
   // up to 65 536: the whole graph in one compute wave next to two I/O waves
   if (!rq.P && !rq.B && n_samples >= 256 && (rq.U == 0 || rq.U == 8 || rq.U == 16 || rq.U == 32) &&
       !(v.flags & (FZ_VF_STAGE_PACK | FZ_VF_NO_STAGE_PACK | FZ_VF_OUT_F64 | FZ_VF_PREFETCH3 | FZ_VF_STREAM_MAJOR))) {
      uint32_t W = 0;
      if (n_streams <= 16384) W = g.wave_roles(4) ? 4 : g.wave_roles(3) ? 3 : 0;
      if (!W && n_streams <= 32768 && g.wave_roles(2)) W = 2;
      if (W) {
         fz_variant q{1, rq.U, 0, v.flags | (W - 1) << 10 | (W < 4 ? (uint32_t)FZ_VF_IO_WAVE : 0u)};
         return resolve_variant(g, &q, n_streams, n_samples, tile_streams, allow_lockstep);
      }
      // one wave per SIMD (config 2: 65 536 streams): the stage-packed wave next to a loader and a storer.  Round 6, paired bursts on four boards:
      // ahead of the lone stage-packed wave in 8 of 8 comparisons on tiles (+0.3 ... +3.1 %) and 6 of 8 on rows (-0.8 ... +4.4 %, mean +1.6 %:
      // 0.682-0.699 against 0.654-0.701), and the most frugal arrangement sustained (0.541 J per launch: profiles/r06/config2_floor.txt)
      // -- where four tuples fit a workgroup's LDS (and its registers: finalize_variant checks after the build), for graphs the lone wave does not spend all
      // its time on arithmetic with (two boards: cascades of 2 / 4 / 6 stages +5-8 / +5 / +0-4 %, the cascade with a gain behind it +22 %, 40 960 streams +6 %,
      // the oscillator chain level; 8 stages -1 ... -2 %, 10 level: profiles/r06/config2_io_waves_default.txt)
      if (n_streams > 32768 && n_streams <= 65536 && g.n_ops <= 64 && g.wave_roles(1)) {
         fz_variant q{1, rq.U ? rq.U : 16u, 0, v.flags | FZ_VF_IO_WAVE | FZ_VF_IO_WAVE2};
         const Variant r = resolve_variant(g, &q, n_streams, n_samples, tile_streams, allow_lockstep);
         if (r.block == 256) return r;
      }
   }
   // stage packing: one stream per lane, pairs of isomorphic graph segments in one v_pk_* (fz_split.cpp); automatic unless the block is
   // so short that the masked steps at either end would dominate
   if (v.flags & FZ_VF_STAGE_PACK) {
      if (!g.split.ok) fail(FZ_E_UNSUPPORTED, "FZ_VF_STAGE_PACK: the graph is not a series of isomorphic segments");
      if (v.P != 1) fail(FZ_E_INVALID, "FZ_VF_STAGE_PACK needs streams_per_lane == 1");
   } else if (!rq.P && v.P == 1 && g.split.ok && !(v.flags & FZ_VF_NO_STAGE_PACK) && n_samples >= 16u * (g.split.atoms() - 1)) {
      v.flags |= FZ_VF_STAGE_PACK;
   }
   v.flags &= ~(uint32_t)FZ_VF_NO_STAGE_PACK;
   v.block = rq.B ? rq.B : 256;
   if (v.flags & FZ_VF_STREAM_MAJOR) return resolve_stream_major(g, rq, v, n_streams, n_samples);
   // stream-tiled frames, packed lanes, chip oversubscribed: the tiles' rows are walked in lockstep too when a CU-wide workgroup of two streams per
   // lane divides the tile and the graph is light on registers (round 6; three boards, tiles of 8192, 1 M streams: the cascade 0.755-0.766 against
   // 0.748-0.762, the fan-out sum 0.784-0.795 against 0.745-0.760: ahead on every board; the oscillator chain with its 31 coefficients per stream
   // 0.59-0.62 against 0.72-0.73: stays free-running; profiles/r06/tiles_in_lockstep.txt) -- else two free-running workgroups per CU
   if (nothing_asked && tile_streams && v.P == 2 && n_streams >= (1u << 19) && !g.n_lds_slots) {
      if (allow_lockstep >= 2 && tile_streams % 2048 == 0 && n_samples >= kLockstepMinRows && g.n_param < 8 && reg_state <= 16 && !g.typed && g.far_lines.empty() &&
          g.n_in <= 2 && g.n_out <= 2) {
         v.U = 2;
         v.block = 1024;
         v.flags |= FZ_VF_LOCKSTEP | FZ_VF_GRID_SYNC;
      } else {
         v.flags |= FZ_VF_MAX_WG(2);
      }
   }
   if (g.n_lds_slots) {
      // LDS rings: a workgroup's rings must fit the CU's 160 KiB (the vectorised rings pad their rows)
      auto bytes = [&](const Variant& w) { return (uint64_t)ring_plan(g, w).slots * w.block * 4u * w.P; };
      while (bytes(v) > kMaxLdsBytes && !rq.B && v.block > 64) v.block /= 2;
      while (bytes(v) > kMaxLdsBytes && !rq.P && v.P > 1) v.P /= 2;
      if (bytes(v) > kMaxLdsBytes)
         fail(FZ_E_UNSUPPORTED, "delay lines too long for the LDS ring buffers of this build (" + std::to_string(g.n_lds_slots) + " slots)");
   }
   return v;
}

// the streams the (first) lockstep launch of a block covers: all of them -- except for the library's own choice on plain time-major
// frames when time_major_geometry peels a remainder off the end (its launch runs the few-stream kernels)
uint64_t lockstep_streams(const Graph& g, const fz_variant* uv, const Variant& v, uint64_t n_streams, uint32_t tile_streams)
{
   // (tiles and LDS rings walk in lockstep at geometries of their own -- whole tiles, 256-lane workgroups: nothing is peeled off)
   if (uv_has_shape(uv) || !(v.flags & FZ_VF_LOCKSTEP) || (tile_streams && tile_streams < n_streams) || g.n_lds_slots) return n_streams;
   return time_major_geometry(n_streams, v.P, g.n_ops > 30, !g.typed && g.far_lines.empty()).main_streams;
}

// The kernel of the REMAINDER launch (the last `rem` streams of a plain time-major block whose laps cover whole workgroups only): one-wave
// workgroups of the ordinary frame kernel, one stream per lane, stage-packed where the graph allows: ~100 registers per lane, so that a
// wave of it fits a SIMD NEXT TO the four of a lap's workgroup (a fatter kernel would keep a lap's workgroup off its CU until the
// remainder is done); a spilling kernel steps down as always.  ONE function for the launch path, fz_program_build_for and
// fz_program_kernel_resources.  Its rows are off the grid by construction: the same store policy rule as the main kernel's.
Variant remainder_variant(fz_program* p, const fz_variant* uv, uint64_t n_streams, uint32_t n_samples, uint64_t rem)
{
   const Graph& g = p->g;
   const bool f64 = uv && (uv->flags & FZ_VF_OUT_F64);
   const bool sp_ok = g.split.ok && n_samples >= 16u * (g.split.atoms() - 1);
   // (far reads are prefetched a chunk ahead: the chunk is at most half the youngest ring read -- the cap resolve_variant checks requests against)
   const uint32_t U = g.far_lines.empty() ? 16u : std::min(16u, std::max(1u, g.far_min_read ? g.far_min_read / 2 : 16u));
   const fz_variant rq{1, U, 64, (sp_ok ? (uint32_t)FZ_VF_STAGE_PACK : (uint32_t)FZ_VF_NO_STAGE_PACK) | (f64 ? (uint32_t)FZ_VF_OUT_F64 : 0u)};
   Variant r = resolve_variant(g, &rq, rem, n_samples, 0, 0);
   const uint64_t wmax = std::max<uint64_t>(std::max(g.n_in, g.n_out), 1), out_w = (uint64_t)std::max<uint32_t>(g.n_out, 1) * (f64 ? 2 : 1);
   while (n_streams * std::max(wmax, out_w) * 4u * r.U >= (1ull << 32) && r.U > (ws_parts(r.flags) ? 8u : 1u)) r.U /= 2;   // (a chunk of U rows: one 4 GiB descriptor)
   if ((n_streams * out_w * 4u) % kStoreGridBytes || ((n_streams - rem) * out_w * 4u) % kStoreGridBytes) r.flags |= FZ_VF_ST_MERGE;
   return settle_variant(p, r);
}

// The kernel a launch of this shape runs: the variant resolved for the layout, fitted to the tile size and the 4 GiB chunk limit,
// its unroll lowered until nothing spills.  ONE function for fz_run_block, fz_program_kernel_name, fz_program_build_for and
// fz_program_kernel_resources, so that what is reported and pre-built is what is launched.
// (settle = false: resolved and fitted only -- nothing is built; what the implicit tune uses to ask whether a candidate is at hand)
Variant finalize_variant(fz_program* p, const fz_variant* uv, uint64_t n_streams, uint32_t n_samples, uint32_t tile_streams, bool settle)
{
   const Graph& g = p->g;
   if (tile_streams >= n_streams) tile_streams = 0;
   const bool stream_major = uv && (uv->flags & FZ_VF_STREAM_MAJOR);
   auto fit = [&](Variant v) {
      if (tile_streams) {
         // a workgroup must not straddle tiles: shrink the lane packing / block until it divides
         const bool fixedP = uv && uv->streams_per_lane, fixedB = uv && uv->block_threads;
         while (tile_streams % (v.P * v.block) && !fixedP && v.P > 1) v.P /= 2;
         while (tile_streams % (v.P * v.block) && !fixedB && v.block > 64) v.block /= 2;
         if (tile_streams % (v.P * v.block)) fail(FZ_E_INVALID, "tile_streams must be a multiple of streams_per_lane * block_threads");
      }
      if (!stream_major) {   // a chunk of U rows is addressed through ONE buffer descriptor: it must stay below 4 GiB
         const uint64_t wmax = std::max<uint64_t>(std::max(g.n_in, g.n_out), 1);
         const uint64_t out_w = (uint64_t)std::max<uint32_t>(g.n_out, 1) * ((uv && (uv->flags & FZ_VF_OUT_F64)) ? 2 : 1);
         const uint64_t row_bytes = (tile_streams ? tile_streams : n_streams) * std::max(wmax, out_w) * 4;
         const uint32_t umin = ws_parts(v.flags) ? 8u : 1u;            // (the wave-split kernels run rounds of 8 / 16 / 32 steps)
         while (row_bytes * v.U >= (1ull << 32) && v.U > umin) {
            if (uv && uv->unroll) fail(FZ_E_INVALID, "unroll x row bytes must stay below 4 GiB: lower the unroll or tile the streams");
            v.U /= 2;
         }
         if (row_bytes * v.U >= (1ull << 32)) fail(FZ_E_UNSUPPORTED, "rows too wide for this kernel variant (unroll x row bytes must stay below 4 GiB): tile the streams");
      }
      // (XCD-wide synchronisation needs every workgroup running: with more blocks than the chip holds workgroups the launch path cuts the
      //  block into laps, one launch each: fz_launch.cpp)
      // Rows off the 16-byte grid -- 1 048 577 streams: whole laps of four streams per lane, but every row starts 4 bytes further off the
      // grid -- cost the b128 accesses 12-17 % of their rate; the lane's streams 64 apart instead (dword accesses) cost more: 9.6 ms
      // against 6.5 ms (profiles/r04/rows_off_the_grid.txt)
      v.flags &= ~FZ_VF_RAGGED;
      if ((v.flags & FZ_VF_LOCKSTEP) && !tile_streams && !stream_major && lockstep_streams(p->g, uv, v, n_streams, 0) % v.P) v.flags |= FZ_VF_RAGGED;
      // output rows off the store grid: stores that let L2 merge the sectors neighbouring waves share (see fz_block_kernel.hip.inc)
      v.flags &= ~FZ_VF_ST_MERGE;
      if (!tile_streams && !stream_major) {
         const uint64_t out_row_bytes = n_streams * (uint64_t)std::max<uint32_t>(g.n_out, 1) * ((uv && (uv->flags & FZ_VF_OUT_F64)) ? 8u : 4u);
         if (out_row_bytes % kStoreGridBytes) v.flags |= FZ_VF_ST_MERGE;
      }
      // LANE GROUPS: no access of a lane wider than 16 bytes, so that every memory instruction of a wave covers whole, contiguous sectors.  A
      // lane of P streams whose frames hold w floats per stream (the wider of in and out) takes its streams in groups of 4 / w, the groups
      // 64 x group size apart: typed frames of 8 bytes per stream with four streams per lane -> two PAIRS (the complex one-pole in lockstep:
      // 0.76 against 0.68 of peak with the lane's 32-byte output slice in two half-sector stores); 4-wire frames with two streams per lane ->
      // two SINGLES (profiles/r05/lane_groups.txt)
      v.flags &= ~(FZ_VF_LANE_PAIRS | FZ_VF_LANE_SINGLES);
      {
         const uint32_t out_floats = g.n_out * ((uv && (uv->flags & FZ_VF_OUT_F64)) ? 2u : 1u), w = std::max<uint32_t>(g.n_in, out_floats);
         const uint32_t group = w == 2 ? 2u : w == 4 ? 1u : 0u;
         if (group && v.P >= 2 * group && !stream_major && !ws_parts(v.flags) && !(v.flags & (FZ_VF_RAGGED | FZ_VF_STAGE_PACK)) && g.far_lines.empty() &&
             n_streams % (64u * v.P) == 0 && (!tile_streams || tile_streams % (64u * v.P) == 0))
            v.flags |= group == 2 ? FZ_VF_LANE_PAIRS : FZ_VF_LANE_SINGLES;
      }
      return settle ? settle_variant(p, v) : v;
   };
   const Variant want = resolve_variant(g, uv, n_streams, n_samples, tile_streams);
   Variant v = fit(want);
   if (settle && !uv_has_shape(uv) && ws_parts(v.flags) == 1 && (v.flags & FZ_VF_IO_WAVE2) && v.block < 256) {
      // the library's own choice of two I/O waves next to the compute wave (config 2's shapes) needs four tuples in a workgroup -- one compute
      // wave on every SIMD of the CU: a graph whose compute wave needs more registers than a third of a SIMD's (the 12-stage cascade) settles
      // at two tuples, its compute waves on half of the SIMDs, 0.23 of peak against 0.39-0.40 for the lone stage-packed wave, which it runs instead
      const fz_variant lone{1, 0, 0, FZ_VF_STAGE_PACK};
      v = fit(resolve_variant(g, &lone, n_streams, n_samples, tile_streams));
   }
   if (settle && (v.flags & FZ_VF_LOCKSTEP) && !(uv && (uv->flags & FZ_VF_LOCKSTEP))) {
      // the library's own lockstep choice needs its kernel in the 128 registers of a 1024-lane workgroup with the rows in flight
      // it was chosen for: step down the streams per lane until it fits; a graph that never does runs the ordinary four-wave
      // workgroups (level 0)
      Variant w = want;
      for (uint32_t level = want.P; level > 0;) {
         const auto k = get_kernel(p, v, nullptr);
         if (k->res.scratch_bytes == 0 && v.U >= w.U) break;
         // before giving up streams per lane: the same packing with ONE row per chunk buffer and three buffers needs fewer
         // registers than chunks of two or four rows (the oscillator chain, two streams per lane, 1024 lanes: 114 against 128 + spills)
         if (w.U > 1 && !(w.flags & FZ_VF_STAGE_PACK) && g.far_lines.empty()) {   // (HBM rings: no third buffer)
            Variant one = w;
            one.U = 1;
            one.flags |= FZ_VF_PREFETCH3;
            const Variant f1 = fit(one);
            if (f1.U == 1 && get_kernel(p, f1, nullptr)->res.scratch_bytes == 0) {
               v = f1;
               break;
            }
         }
         // (a graph with many per-stream coefficients that is a series of isomorphic segments goes straight to one stream per lane,
         //  STAGE-PACKED: packing by stages costs no registers per stream, packing by lanes doubles the coefficient registers -- the
         //  oscillator chain with its 31: 6.45 ms against 6.74-6.80 ms with two streams per lane, and 6.60 ms with four in 512-lane
         //  workgroups, on the board that ran all three; ahead on two more boards; profiles/r04/sweep_time_major_geometry.txt)
         level = level == 4 ? ((g.n_param >= 16 && g.split.ok) ? 1 : 2) : level - 1;
         w = resolve_variant(g, uv, n_streams, n_samples, tile_streams, level);
         v = fit(w);
         if (!(v.flags & FZ_VF_LOCKSTEP)) break;
      }
   }
   return v;
}

}  // namespace fz
