// inflate_seg3.h -- landing decoder: the counting pass of the interval decoder (inflate_seg2.h) built
// again around wavefront-uniform control flow (round 5).
//
// Same contract as seg2_plan: one ultra-fast-format stream (reference src/compress/ultrafast.rs:82-181;
// inner loop src/decompress.rs:645-830) per wavefront, its block data cut into 64 equal bit segments,
// a GUESSED chain per lane through a window at the start of its segment, every byte counted, the
// chain of every lane cut into INTERVALS of at most kS2Meter look-ups that end behind their run chain,
// checkpoints (bit position, bytes so far) in the wavefront's scratch, a plan for seg2_write.
// What is different is how it gets there (tests/seg3_model.py is the executable statement of the rules):
//   * ONE counted chain per lane and an exact LANDING instead of two walks through the window: lane l
//     counts from x0[l] -- where its guessed chain left its window (lane 0: the first token) -- straight
//     through the next lane's window and must end exactly on x0[l + 1]: whole periods of 32 look-ups
//     while they cannot pass it, then groups of 8, pairs, single look-ups, and the first literal of a
//     step alone once the whole step would pass (both tables are in the LDS: the step table of
//     inflate_seg2_groups.h and the reference-layout table of inflate_tables.h, which also decodes the
//     run / end-of-block tokens without a canonical walk).  A lane that lands proves its right
//     neighbour's guess, by induction from lane 0; a lane that cannot leaves the stream to the interval
//     kernel behind this one.
//   * the lanes move in lockstep: every 32 look-ups one event for all of them -- 64 B per lane from
//     global memory into a ring of 32 dwords per lane (requested a period ahead, 16 dwords at a time,
//     two ds_write2st64 per four dwords), one checkpoint store of 512 contiguous bytes -- instead of
//     per-lane ring levels, predicates and pending chunks at every 16 look-ups.
//   * a lane that meets a token that is no literal marks time to the end of its period (a zero entry of
//     the step table changes nothing) and all such lanes take their run chains together.
#pragma once
#include "inflate_seg2.h"

namespace fdh {

constexpr uint32_t kS3MinDataBits = 64 * 1024;  // shorter streams are left to the interval kernel (its segments adapt)
constexpr uint32_t kS3Window = 256;             // bits of a segment the guessed chain walks before it is believed
constexpr uint32_t kS3RingWords = 32;           // input ring, dwords per lane ([word][lane] layout: conflict-free)
constexpr uint32_t kS3PeriodPairs = 16;         // pairs of look-ups between two events
constexpr uint32_t kS3PeriodBits = 2 * kS3PeriodPairs * kLitBits;  // most stream bits a period consumes
constexpr uint32_t kS3GroupBits = 2 * kS2Pairs * kLitBits;
constexpr uint32_t kS3TailGuard = 256;          // bytes behind the end of a stream that a lane may load
constexpr uint32_t kS3InCap = 3072;             // input image of the writing pass in this kernel's LDS layout

struct Seg3Lds {
    uint32_t lit[kLitSize];        // step table (seg2_entry_build), at LDS offset 0
    uint32_t canon[kLitSize];      // reference-layout table (inflate_tables.h)
    uint32_t w[kS2Waves][2048];    // per wavefront, 8 KiB aligned: the ring / the two images of the writing pass
};
static_assert(sizeof(Seg3Lds) == 160 * 1024, "one workgroup owns the LDS of its CU");

#ifdef FDH_S3_DEBUG
__device__ uint32_t g_s3stat[16];
__device__ uint32_t g_s3time[4096 * 16];
#define S3STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_s3stat[k], (uint32_t)(v)); } while (0)
#define S3T(k) do { if (sid < 4096 && (threadIdx.x & 63) == 0) g_s3time[sid * 16 + (k)] = (uint32_t)clock64(); } while (0)
#define S3N(k, v) do { if (sid < 4096 && (threadIdx.x & 63) == 0) g_s3time[sid * 16 + (k)] = (uint32_t)(v); } while (0)
#else
#define S3STAT(k, v) do { } while (0)
#define S3T(k) do { } while (0)
#define S3N(k, v) do { } while (0)
#endif

// seg2_count_group on a ring of 32 words per lane (word w of a lane at rb + (w & 31) * 256).
__device__ __forceinline__ uint32_t seg3_count_group(uint32_t pairs, uint32_t rb, uint32_t& lo, uint32_t& hi, uint32_t& c,
                                                     uint32_t& ra) {
    uint32_t e, t, nw;
    const uint32_t k256 = 256u, m1f00 = 0x1f00u;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 " S2_WLO ", %[lo]\n"
        "  v_mov_b32 " S2_WHI ", %[hi]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "Lpair_%=:\n"
        "  v_lshrrev_b64 " S2_SHF ", %[c], " S2_WIN "\n"
        "  v_and_b32 %[t], 0x3ffc, " S2_SH0 "\n"
        "  ds_read_b32 %[e], %[t]\n"
        "  s_waitcnt lgkmcnt(0)\n"
        S2_ADD_BYTE0("%[c]", "%[e]")
        "  v_lshrrev_b64 " S2_SHF ", %[c], " S2_WIN "\n"
        "  v_and_b32 %[t], 0x3ffc, " S2_SH0 "\n"
        "  ds_read_b32 %[e], %[t]\n"
        "  s_waitcnt lgkmcnt(0)\n"
        S2_ADD_BYTE0("%[c]", "%[e]")
        "  v_and_b32 %[t], 32, %[c]\n"
        "  v_cmp_ne_u32 vcc, 0, %[t]\n"
        "  v_and_b32 %[c], 0xffffffdf, %[c]\n"
        "  s_sub_u32 %[pairs], %[pairs], 1\n"
        "  v_cndmask_b32 " S2_WLO ", " S2_WLO ", " S2_WHI ", vcc\n"
        "  v_cndmask_b32 " S2_WHI ", " S2_WHI ", %[nw], vcc\n"
        "  v_cndmask_b32 %[t], 0, %[k256], vcc\n"
        "  v_add_u32 %[t], %[ra], %[t]\n"
        "  v_and_or_b32 %[ra], %[t], %[m1f00], %[rb]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "  s_cmp_lg_u32 %[pairs], 0\n"
        "  s_cbranch_scc1 Lpair_%=\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 %[lo], " S2_WLO "\n"
        "  v_mov_b32 %[hi], " S2_WHI "\n"
        : [pairs] "+s"(pairs), [lo] "+v"(lo), [hi] "+v"(hi), [c] "+v"(c), [ra] "+v"(ra), [e] "=&v"(e), [t] "=&v"(t),
          [nw] "=&v"(nw)
        : [rb] "v"(rb), [k256] "v"(k256), [m1f00] "s"(m1f00)
        : "vcc", "scc", "memory", S2_CLOBBER4);
    return e;
}

// The token at the read position that is no literal step, from the reference-layout table: a run (length
// symbol, extra bits, the one distance code of the prefix: '0' = distance 1, src/decompress.rs:793-801),
// the end-of-block code, or nothing valid.  `w` = 30 stream bits from the token on.
struct S3Tok {
    uint32_t used, run, len1;
    bool eob, bad;
};
__device__ __forceinline__ S3Tok s3_token(const uint32_t* canon, uint32_t w) {
    const uint32_t ce = canon[w & (kLitSize - 1)];
    const uint32_t kind = (ce >> 4) & 15, nb = ce & 15;
    S3Tok t;
    const bool is_len = kind == K_LEN;
    const uint32_t ex = (ce >> 8) & 31;
    t.run = is_len ? (ce >> 16) + ((w >> nb) & ((1u << ex) - 1)) : 0u;
    t.used = is_len ? nb + ex + 1 : nb;
    t.eob = kind == K_EOB;
    t.bad = !t.eob && (!is_len || ((w >> (nb + ex)) & 1) != 0);
    t.len1 = ce >> 24;  // K_LIT1 / K_LIT2: bits of the first literal
    return t;
}

// Counting pass of one stream.  False: not for this kernel, or left PENDING (lane 0 has listed it).
__device__ __forceinline__ bool seg3_plan(const SegArgs& a, const uint32_t* lit, const uint32_t* canon, uint32_t* ring, uint2* ckpt,
                                          const uint64_t sid, S2Plan& plan) {
    const int lane = threadIdx.x & (kWave - 1);
    if (sid >= a.n) return false;

    // ---- stream set-up (uniform) ----
    S3T(0);
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
    const uint8_t* in = a.in + i0;
    const uint64_t ilen = i1 - i0, ocap = o1 - o0;
    const uint8_t* const buf_hi = a.in + a.in_off[a.n];
    bool ours = ilen < (1ull << 19) && ocap < (1ull << 24) && ilen * 8 >= (uint64_t)a.canon_bits + kS3MinDataBits &&
                in + ilen + kS3TailGuard <= buf_hi;
    if (ours) {  // canonical prefix: lane k compares stream dword k
        bool mismatch = false;
        if (lane < 14) {
            uint32_t v = 0;
            const uint8_t* p = in + 4 * lane;
            for (int k = 0; k < 4; k++) v |= (uint32_t)p[k] << (8 * k);
            if (lane == 13) v &= (1u << (a.canon_bits - 13 * 32)) - 1;
            mismatch = v != a.canon_hdr[lane];
        }
        ours = !__any(mismatch);
    }
    if (!ours) {
        if (lane == 0) seg_leave_pending(a, sid);
        return false;
    }
    S3T(1);
    const uint32_t in_bits = (uint32_t)(ilen * 8);
    const uint32_t cap = (uint32_t)ocap;
    const uint32_t data_bits = in_bits - a.canon_bits;
    const uint32_t seg = (data_bits + kWave - 1) / kWave;
    const uint32_t seg_bit0 = a.canon_bits + (uint32_t)lane * seg;

    // ---- the reader: a ring of 32 dwords per lane, the window two bits in front of the next token ----
    const uint32_t rb = lds_offset(ring) + 4 * (uint32_t)lane;
    uint32_t* const rl = ring + lane;  // word w of this lane: rl[(w & 31) * 64]
    const uint32_t wbit = seg_bit0 - 2;
    const uint8_t* gp;
    uint32_t Rw, Ww, lo, hi, c;
    {
        const uint8_t* addr = in + (wbit >> 3);
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(addr) & 15);
        gp = addr - mis;
        Rw = mis >> 2;
        c = 8 * (mis & 3) + (wbit & 7);
    }
    const uint32_t base_tok = 8 * (uint32_t)(gp - in) + 2;  // token position = base_tok + 32 Rw + (c & 63)
    {
        uint4 q[8];
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = reinterpret_cast<const uint4*>(gp)[k];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            rl[(4 * k + 0) * 64] = q[k].x;
            rl[(4 * k + 1) * 64] = q[k].y;
            rl[(4 * k + 2) * 64] = q[k].z;
            rl[(4 * k + 3) * 64] = q[k].w;
        }
        gp += 128;
        Ww = 32;
    }
    uint4 pd0, pd1, pd2, pd3;  // the 64 B requested a period ago
    pd0 = reinterpret_cast<const uint4*>(gp)[0];
    pd1 = reinterpret_cast<const uint4*>(gp)[1];
    pd2 = reinterpret_cast<const uint4*>(gp)[2];
    pd3 = reinterpret_cast<const uint4*>(gp)[3];
    gp += 64;
    wave_sync();
    lo = rl[(Rw & 31) * 64];
    hi = rl[((Rw + 1) & 31) * 64];

    auto position = [&]() __attribute__((always_inline)) { return base_tok + 32 * Rw + (c & 63u); };
    auto group = [&](uint32_t pairs) __attribute__((always_inline)) {
        uint32_t ra = rb + (((Rw + 2) & 31u) << 8);
        const uint32_t ra0 = ra;
        const uint32_t e = seg3_count_group(pairs, rb, lo, hi, c, ra);
        Rw += ((ra - ra0) >> 8) & 31u;
        return e;
    };
    auto window30 = [&]() __attribute__((always_inline)) { return __builtin_amdgcn_alignbit(hi, lo, c & 63u) >> 2; };
    auto advance = [&](uint32_t bits) __attribute__((always_inline)) {  // bits <= 32, per lane
        c += bits;
        const bool wrap = (c & 63u) >= 32;
        const uint32_t nw = rl[((Rw + 2) & 31) * 64];
        lo = wrap ? hi : lo;
        hi = wrap ? nw : hi;
        Rw += wrap ? 1u : 0u;
        c -= wrap ? 32u : 0u;
    };

    S3T(2);
    // ---- guessed chain through the window (lanes 1..63); a run token is stepped over, anything else
    //      that is no literal slides on by one bit ----
    uint32_t pos = position();
    {
        const uint32_t leave = seg_bit0 + kS3Window;
        for (int iter = 0; iter < 64; iter++) {
            const bool act = lane > 0 && pos < leave;
            if (!__any(act)) break;
            uint32_t e = 1;
            if (act) e = group(kS2Pairs);
            pos = position();
            const bool parked = act && e == 0 && pos < leave;
            if (__any(parked)) {
                const S3Tok t = s3_token(canon, window30());
                advance(parked ? ((t.run != 0 && !t.bad) ? t.used : 1u) : 0u);
                pos = position();
            }
        }
    }
    S3T(3);
    bool fault = lane > 0 && pos < seg_bit0 + kS3Window;  // (64 groups did not leave the window: cannot happen)
    const uint32_t x0 = pos;
    const uint32_t nxt = __shfl_down(x0, 1, kWave);
    const uint32_t target = lane == kWave - 1 ? in_bits : nxt;

    // ---- the counted chain: from x0 to exactly `target`; checkpoints in the lane's column of the scratch ----
    uint2* const ckrow = ckpt + (uint32_t)lane;
    uint32_t slot = kS2HeadSlots, m = 0, bl = 0, eob_bits = 0;
    bool stopped = false;
    c &= 63u;
    auto cut = [&](bool doit) __attribute__((always_inline)) {
        if (doit) {
            if (slot >= kS2Slots) {
                fault = true;
            } else {
                const uint32_t cnt = c >> 6;
                ckrow[slot * S2_CK_STRIDE] = make_uint2((pos - seg_bit0) | (bl << kS2PosBits), cnt - 16 * bl);
                slot++;
            }
            m = 0;
        }
    };
    // event: what was requested a period ago goes into the ring where there is room for it (16 dwords), and the
    // next 64 B are requested.  After it a lane has at least 16 dwords in front of it: a period's 12 + the window.
    auto refill = [&]() __attribute__((always_inline)) {
        const bool put = Ww - Rw <= 16;
        if (put) {
            uint32_t* const p = rl + (Ww & 16u) * 64;
            p[0 * 64] = pd0.x; p[1 * 64] = pd0.y; p[2 * 64] = pd0.z; p[3 * 64] = pd0.w;
            p[4 * 64] = pd1.x; p[5 * 64] = pd1.y; p[6 * 64] = pd1.z; p[7 * 64] = pd1.w;
            p[8 * 64] = pd2.x; p[9 * 64] = pd2.y; p[10 * 64] = pd2.z; p[11 * 64] = pd2.w;
            p[12 * 64] = pd3.x; p[13 * 64] = pd3.y; p[14 * 64] = pd3.z; p[15 * 64] = pd3.w;
            Ww += 16;
            pd0 = reinterpret_cast<const uint4*>(gp)[0];
            pd1 = reinterpret_cast<const uint4*>(gp)[1];
            pd2 = reinterpret_cast<const uint4*>(gp)[2];
            pd3 = reinterpret_cast<const uint4*>(gp)[3];
            gp += 64;
        }
    };
    // the run chains of the lanes in `mask` (they sit on a token that is no literal); a chain ends its interval
    auto special = [&](bool mask) __attribute__((always_inline)) {
        refill();  // (a period may have used up what the last event guaranteed; the chain reads on)
        bool go = mask;
        uint32_t chain = 0;
        for (int rep = 0; rep < kS2Repeat && __any(go); rep++) {
            const S3Tok t = s3_token(canon, window30());
            const bool is_run = go && t.run != 0 && !t.bad && pos + t.used <= target;
            const bool is_eob = go && rep == 0 && t.eob && lane == kWave - 1;
            if (go && rep == 0 && !is_run && !is_eob) fault = true;
            if (is_eob) {
                stopped = true;
                eob_bits = t.used;
            }
            chain += is_run ? t.run : 0u;
            advance(is_run ? t.used : 0u);
            pos = position();
            go = is_run && t.run == 258 && pos < target;
        }
        c += chain << 6;
        bl += chain >= kS2LongRun ? chain / 16 - 1 : 0u;
        cut(mask && chain != 0);
    };
    cut(true);
    uint32_t dbg_periods = 0, dbg_special = 0;
    (void)dbg_periods;
    (void)dbg_special;
    // whole periods
    for (;;) {
        refill();
        const bool bulk = !stopped && !fault && pos + kS3PeriodBits <= target;
        if (!__any(bulk)) break;
        cut(bulk && m > 0);
        uint32_t e = 1;
        if (bulk) {
            m = 2 * kS3PeriodPairs;
            e = group(kS3PeriodPairs);
        }
        pos = position();
        const bool parked = bulk && e == 0;
        dbg_periods++;
        if (__any(parked)) {
            dbg_special++;
            special(parked);
        }
    }
    S3T(4);
    S3N(10, dbg_periods);
    S3N(11, dbg_special);
    // groups, pairs
#pragma unroll
    for (int stage = 0; stage < 2; stage++) {
        const uint32_t pairs = stage == 0 ? kS2Pairs : 1u;
        const uint32_t bits = 2 * pairs * kLitBits;
        for (;;) {
            const bool act = !stopped && !fault && pos + bits <= target;
            if (!__any(act)) break;
            cut(act && m + 2 * pairs > kS2Meter);
            uint32_t e = 1;
            if (act) {
                m += 2 * pairs;
                e = group(pairs);
            }
            pos = position();
            const bool parked = act && e == 0;
            if (__any(parked)) special(parked);
        }
    }
    S3T(5);
    // single steps; the first literal alone once the whole step would pass the target
    for (int iter = 0; iter < 64; iter++) {
        const bool act = !stopped && !fault && pos < target;
        if (!__any(act)) break;
        const uint32_t w = window30();
        const uint32_t e = lit[w & (kLitSize - 1)];
        const bool spec = act && e == 0;
        const bool step = act && e != 0;
        cut(step && m + 1 > kS2Meter);
        m += step ? 1u : 0u;
        const uint32_t used = e & 15u, d = target - pos;
        const uint32_t len1 = canon[w & (kLitSize - 1)] >> 24;
        const bool full = step && used <= d;
        const bool one = step && !full && len1 <= d;
        if (step && !full && !one) fault = true;  // no token ends on the target: the neighbour's guess was wrong
        c += (full ? (e >> 6) & 3u : (one ? 1u : 0u)) << 6;
        advance(full ? used : (one ? len1 : 0u));
        pos = position();
        if (__any(spec)) special(spec);
    }
    S3T(6);
    const bool landed = !fault && (lane == kWave - 1 ? stopped : (!stopped && pos == target));
    cut(m > 0 || slot == kS2HeadSlots + 1);  // the end of the chain is a checkpoint too (unless a cut just made it one)

    // ---- the plan ----
    bool ok = !__any(!landed) && !__any(fault);
    const uint32_t count = c >> 6;
    const uint32_t n_int = slot - kS2HeadSlots - 1;
    ok = ok && !__any(slot < kS2HeadSlots + 2);
    unsigned long long incl = count;
    uint32_t incl_b = bl, incl_n = n_int;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const unsigned long long y = __shfl_up(incl, o, kWave);
        const uint32_t yb = __shfl_up(incl_b, o, kWave), yn = __shfl_up(incl_n, o, kWave);
        if (lane >= o) {
            incl += y;
            incl_b += yb;
            incl_n += yn;
        }
    }
    const unsigned long long total64 = ((unsigned long long)__builtin_amdgcn_readlane((uint32_t)(incl >> 32), kWave - 1) << 32) |
                                       __builtin_amdgcn_readlane((uint32_t)incl, kWave - 1);
    ok = ok && total64 <= cap;
    ok = ok && __builtin_amdgcn_readlane(incl_b, kWave - 1) < (1u << (32 - kS2PosBits));
    const uint32_t eob_end = __builtin_amdgcn_readlane(pos + eob_bits, kWave - 1);
    const uint32_t tb = (eob_end + 7) >> 3;
    ok = ok && (uint64_t)tb * 8 + 32 <= in_bits;
    S3STAT(0, 1);
    S3STAT(1, ok ? 0 : 1);
    S3STAT(2, __popcll(__ballot(!landed)));
    if (!ok) {
        if (lane == 0) seg_leave_pending(a, sid);
        return false;
    }
    plan.n_int = n_int;
    plan.hn = 0;
    plan.P = incl_n - n_int;
    plan.obase = (uint32_t)incl - count;
    plan.bbase = incl_b - bl;
    plan.hc = 0;
    plan.hb = 0;
    plan.total = (uint32_t)total64;
    plan.ni = __builtin_amdgcn_readlane(incl_n, kWave - 1);
    plan.tb = tb;
    plan.seg = seg;
    S3T(7);
    return true;
}

// False: the stream was passed on (too short / not canonical / near the end of the batch buffer, or
// after a counting pass that did not land).
__device__ __forceinline__ bool seg3_decode(const SegArgs& a, Seg3Lds& L, uint2* ckpt, const uint64_t sid) {
    const uint32_t wid = threadIdx.x / kWave;
    uint32_t* const W = L.w[wid];
    S2Plan plan;
    const bool planned = seg3_plan(a, L.lit, L.canon, W, ckpt, sid, plan);
    // the writing pass of the interval decoder: output image in the first 5 KiB, input image in the last 3
    if (planned && (a.flags & 0x40000u)) {  // debug (FDH_FLAG_LANDING_COUNT_ONLY): time the counting pass alone
        if ((threadIdx.x & (kWave - 1)) == 0) seg_leave_pending(a, sid);
        return true;
    }
    if (planned) seg2_write<kS3InCap>(a, L.lit, W + kS2BWords, W, ckpt, sid, plan);
    S3T(8);
    return planned;
}

}  // namespace fdh
