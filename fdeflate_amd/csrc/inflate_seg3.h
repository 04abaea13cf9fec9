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
// Round 6: a stream the lean writer (seg3_write) will take -- known at set-up -- merges up to kS3Repeat run tokens into
// a chain (five 258-byte tokens per step in a flat stretch) and marks the intervals that are a chain behind a chain
// (kS3PureFlag): such chains never enter the writer's image, whose zeros they would only leave as they are; the
// writer stores them straight to the slot.  The tail of a chain is taken in halves (8, 4, 2, 1 pairs of look-ups).
#pragma once
#include "inflate_seg2.h"

namespace fdh {

constexpr uint32_t kS3MinSegBits = 1024;        // no segment shorter than this: a short stream uses fewer lanes
constexpr uint32_t kS3Window = 256;             // bits of a segment the guessed chain walks before it is believed
constexpr uint32_t kS3RingWords = 32;           // input ring, dwords per lane ([word][lane] layout: conflict-free)
constexpr uint32_t kS3PeriodPairs = 16;         // pairs of look-ups between two events
constexpr uint32_t kS3PeriodBits = 2 * kS3PeriodPairs * kLitBits;  // most stream bits a period consumes
constexpr uint32_t kS3GroupBits = 2 * kS2Pairs * kLitBits;
constexpr uint32_t kS3TailGuard = 256;          // bytes behind the end of a stream that a lane may load
constexpr uint32_t kS3InCap = 3072;             // input image of the writing pass in this kernel's LDS layout
constexpr int kS3Repeat = 64;                   // run tokens merged into one chain when the lean writer takes the stream
constexpr uint32_t kS3FlatBits = 10;            // bits of a 258-byte run token of the prefix's code (symbol 285 + distance '0')
constexpr uint32_t kS3PureFlag = 0x80000000u;   // checkpoint.y: the interval that starts here is a run chain behind a run chain

struct Seg3Lds {
    uint32_t lit[kLitSize];        // step table (seg2_entry_build), at LDS offset 0
    uint32_t canon[kLitSize];      // reference-layout table (inflate_tables.h)
    uint32_t w[kS2Waves][2048];    // per wavefront, 8 KiB aligned: the ring / the two images of the writing pass
};
static_assert(sizeof(Seg3Lds) == 160 * 1024, "one workgroup owns the LDS of its CU");

#ifdef FDH_S3_DEBUG
__device__ uint32_t g_s3stat[16];
__device__ uint32_t g_s3time[4096 * 16];
__device__ uint32_t g_s3base;  // the 4 096 streams from this one on are sampled (fdh_debug_s3base)
#define S3SMP (sid - g_s3base < 4096)
#define S3IDX ((uint32_t)(sid - g_s3base))
#define S3STAT(k, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_s3stat[k], (uint32_t)(v)); } while (0)
#define S3T(k) do { if (S3SMP && (threadIdx.x & 63) == 0) g_s3time[S3IDX * 16 + (k)] = (uint32_t)clock64(); } while (0)
#define S3N(k, v) do { if (S3SMP && (threadIdx.x & 63) == 0) g_s3time[S3IDX * 16 + (k)] = (uint32_t)(v); } while (0)
__device__ uint32_t g_s3wtime[4096 * 8];
#define S3W_DECL uint32_t s3w_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long s3wt_ = clock64()
#define S3W(k) do { const long long n_ = clock64(); s3w_[k] += (uint32_t)(n_ - s3wt_); s3wt_ = n_; } while (0)
#define S3W_OUT do { if (S3SMP && (threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; k_++) g_s3wtime[S3IDX * 8 + k_] = s3w_[k_]; } while (0)
#else
#define S3W_DECL do { } while (0)
#define S3W(k) do { } while (0)
#define S3W_OUT do { } while (0)
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
// `lean`: the stream qualifies for seg3_write -- no run chain leaves lines out of the image, the slot is 16-B aligned,
// the stream does not end within an input image of the end of the batch buffer.
__device__ __forceinline__ bool seg3_plan(const SegArgs& a, const uint32_t* lit, const uint32_t* canon, uint32_t* ring, uint2* ckpt,
                                          const uint64_t sid, S2Plan& plan, bool& lean) {
    const int lane = threadIdx.x & (kWave - 1);
    if (sid >= a.n) return false;

    // ---- stream set-up (uniform) ----
    S3T(0);
    const uint64_t i0 = a.in_off[sid], i1 = a.in_off[sid + 1];
    const uint64_t o0 = a.out_off[sid], o1 = a.out_off[sid + 1];
    const uint8_t* in = a.in + i0;
    const uint64_t ilen = i1 - i0, ocap = o1 - o0;
    const uint8_t* const buf_hi = a.in + a.in_off[a.n];
    bool ours = ilen < (1ull << 19) && ocap < (1ull << 24) && ilen * 8 >= (uint64_t)a.canon_bits + 44;
    // (uniform) a lane may load kS3TailGuard bytes past the end of its stream: the last stream(s) of the batch take the
    // range-checked loads
    const bool edge = in + ilen + kS3TailGuard > buf_hi;
    // (uniform) the lean writer takes the stream: 16-B aligned slot, not within an input image of the end of the batch
    // buffer.  Known here, because the counting pass may then merge up to kS3Repeat run tokens into a chain (the
    // general writer: kS2Repeat) and mark the chains that follow a chain.
    lean = !(a.flags & 0x80000u) && ((reinterpret_cast<uintptr_t>(a.out) + o0) & 15) == 0 && in + ilen + kS3InCap + 128 <= buf_hi;
    const int reps = lean ? kS3Repeat : kS2Repeat;
    const uint32_t pure_flag = lean ? kS3PureFlag : 0u;
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
    const uint32_t nseg = min((uint32_t)kWave, max(1u, data_bits / kS3MinSegBits));
    const uint32_t seg = (data_bits + nseg - 1) / nseg;
    const int last = (int)nseg - 1;                       // the lane whose chain ends on the end-of-block code
    const bool in_range = (uint32_t)lane < nseg;          // (the other lanes idle: their reader sits on segment 0)
    const uint32_t seg_bit0 = a.canon_bits + (in_range ? (uint32_t)lane * seg : 0u);

    // ---- the reader: a ring of 32 dwords per lane, the window two bits in front of the next token ----
    const uint32_t rb = lds_offset(ring) + 4 * (uint32_t)lane;
    uint32_t* const rl = ring + lane;  // word w of this lane: rl[(w & 31) * 64]
    const uint8_t* gp;
    uint32_t Rw, Ww, lo, hi, c, base_tok;  // token position = base_tok + 32 Rw + (c & 63)
    uint4 pd0, pd1, pd2, pd3;              // the 64 B requested a period ago
    auto ld16 = [&](const uint8_t* p) __attribute__((always_inline)) {
        if (!edge) return *reinterpret_cast<const uint4*>(p);
        const SegChunk ch = seg_load(p, a.in, buf_hi);
        return make_uint4(ch.w[0], ch.w[1], ch.w[2], ch.w[3]);
    };
    // puts the reader of the lanes in `mk` on the token at stream bit `tok`: 128 B into the ring, 64 B requested
    auto start_reader = [&](bool mk, uint32_t tok) __attribute__((always_inline)) {
        if (mk) {
            const uint32_t wbit = tok - 2;
            const uint8_t* addr = in + (wbit >> 3);
            const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(addr) & 15);
            gp = addr - mis;
            Rw = mis >> 2;
            c = 8 * (mis & 3) + (wbit & 7);
            base_tok = 8 * (uint32_t)(gp - in) + 2;
            uint4 q[8];
#pragma unroll
            for (int k = 0; k < 8; k++) q[k] = ld16(gp + 16 * k);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                rl[(4 * k + 0) * 64] = q[k].x;
                rl[(4 * k + 1) * 64] = q[k].y;
                rl[(4 * k + 2) * 64] = q[k].z;
                rl[(4 * k + 3) * 64] = q[k].w;
            }
            gp += 128;
            Ww = 32;
            pd0 = ld16(gp);
            pd1 = ld16(gp + 16);
            pd2 = ld16(gp + 32);
            pd3 = ld16(gp + 48);
            gp += 64;
        }
        wave_sync();
        if (mk) {
            lo = rl[(Rw & 31) * 64];
            hi = rl[((Rw + 1) & 31) * 64];
        }
    };
    start_reader(true, seg_bit0);

    auto position = [&]() __attribute__((always_inline)) { return base_tok + 32 * Rw + (c & 63u); };
    auto group = [&](uint32_t pairs) __attribute__((always_inline)) {
        uint32_t ra = rb + (((Rw + 2) & 31u) << 8);
        const uint32_t ra0 = ra;
#ifdef FDH_X_DUAL
        const uint32_t e = seg3_count_group_x2(pairs, rb, lo, hi, c, ra);
#else
        const uint32_t e = seg3_count_group(pairs, rb, lo, hi, c, ra);
#endif
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

    // A flat stretch: how many times (0..5) the 258-byte run token `tok10` is there again at the read position -- 50 stream
    // bits compared with the token's own ten bits repeated (the two ring words behind the window must be there).
    auto flat_more = [&](uint32_t tok10) __attribute__((always_inline)) {
        const uint32_t o = (c & 63u) + 2;  // (2 <= o <= 33)
        const unsigned long long A = ((unsigned long long)hi << 32) | lo;
        const unsigned long long B = ((unsigned long long)rl[((Rw + 3) & 31) * 64] << 32) | rl[((Rw + 2) & 31) * 64];
        const unsigned long long v = (A >> o) | (B << (64 - o));
        const unsigned long long five = (unsigned long long)tok10 * ((1ull << 0) | (1ull << 10) | (1ull << 20) | (1ull << 30) | (1ull << 40));
        const unsigned long long diff = ((v ^ five) & ((1ull << 50) - 1)) | (1ull << 50);
        return (uint32_t)__builtin_ctzll(diff) / kS3FlatBits;
    };
    auto advance_flat = [&](uint32_t k) __attribute__((always_inline)) {  // k <= 5 tokens of kS3FlatBits bits
        const uint32_t k1 = min(k, 3u);
        advance(kS3FlatBits * k1);
        advance(kS3FlatBits * (k - k1));
    };

    S3T(2);
    // ---- guessed chain through the window (lanes 1..63); a run token is stepped over, anything else
    //      that is no literal slides on by one bit ----
    uint32_t pos = position();
    {
        const uint32_t leave = seg_bit0 + kS3Window;
        for (int iter = 0; iter < 64; iter++) {
            const bool act = in_range && lane > 0 && pos < leave;
            if (!__any(act)) break;
            uint32_t e = 1;
            if (act) e = group(kS2Pairs);
            pos = position();
            const bool parked = act && e == 0 && pos < leave;
            if (__any(parked)) {
                const uint32_t w30 = window30();
                const S3Tok t = s3_token(canon, w30);
                advance(parked ? ((t.run != 0 && !t.bad) ? t.used : 1u) : 0u);
                pos = position();
                const bool flat = parked && t.run == 258 && !t.bad && t.used == kS3FlatBits && pos < leave;
                if (__any(flat)) {  // a flat stretch: up to five more of the same token at once
                    const uint32_t k = flat ? min(flat_more(w30 & ((1u << kS3FlatBits) - 1)), (leave - pos + kS3FlatBits - 1) / kS3FlatBits) : 0u;
                    advance_flat(k);
                    pos = position();
                }
            }
        }
    }
    S3T(3);
    bool fault = in_range && lane > 0 && pos < seg_bit0 + kS3Window;  // (64 groups did not leave the window: cannot happen)
    uint32_t start = pos;   // where the lane's counted chain starts (round 0: where its guessed chain left the window)
    uint32_t target = 0;    // ... and where it must end: the start of the lane to its right

    // ---- the counted chain: from x0 to exactly `target`; checkpoints in the lane's column of the scratch ----
    uint2* const ckrow = ckpt + (uint32_t)lane;
    uint32_t slot = kS2HeadSlots, m = 0, bl = 0, eob_bits = 0;
    bool stopped = !in_range;
    bool over = false;    // the chain has passed the target: no token ends there, the right neighbour's guess was wrong
    bool landed = !in_range;
    bool need = in_range;  // lanes that count in this round (the first: all; then the right neighbours of chains that went over)
    auto cut = [&](bool doit, uint32_t flag = 0u) __attribute__((always_inline)) {
        if (doit) {
            if (slot >= kS2Slots) {
                fault = true;
            } else {
                const uint32_t cnt = c >> 6;
                ckrow[slot * S2_CK_STRIDE] = make_uint2((pos - seg_bit0) | (bl << kS2PosBits), (cnt - 16 * bl) | flag);
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
            pd0 = ld16(gp);
            pd1 = ld16(gp + 16);
            pd2 = ld16(gp + 32);
            pd3 = ld16(gp + 48);
            gp += 64;
        }
    };
    // the run chains of the lanes in `mask` (they sit on a token that is no literal); a chain ends its interval
    auto special = [&](bool mask) __attribute__((always_inline)) {
        refill();  // (a period may have used up what the last event guaranteed; the chain reads on)
        // chain after chain while a lane sits on run tokens (a flat stretch is nothing else): each one its own interval
        bool more = mask;
        for (int nch = 0; nch < 48 && __any(more); nch++) {
            bool go = more;
            uint32_t chain = 0;
            for (int rep = 0; rep < reps && __any(go);) {
                const uint32_t w30 = window30();
                const S3Tok t = s3_token(canon, w30);
                const bool is_run = go && t.run != 0 && !t.bad;
                const bool is_eob = go && rep == 0 && t.eob && lane == last;
                over = over || (is_run && pos + t.used > target);
                if (go && rep == 0 && nch == 0 && !is_run && !is_eob) fault = true;
                if (is_eob) {
                    stopped = true;
                    eob_bits = t.used;
                }
                chain += is_run ? t.run : 0u;
                advance(is_run ? t.used : 0u);
                pos = position();
                go = is_run && t.run == 258 && pos < target;
                rep++;
                // a flat stretch: the same 258-byte token again and again -- up to five more of them at once, by
                // comparing 50 stream bits with the token's own bits repeated (lean streams only: the chains get long)
                const bool flat = lean && go && t.used == kS3FlatBits;
                if (__any(flat) && rep + 5 <= reps) {
                    refill();
                    uint32_t k = flat ? flat_more(w30 & ((1u << kS3FlatBits) - 1)) : 0u;      // tokens that repeat (0..5)
                    k = min(k, (target - pos + kS3FlatBits - 1) / kS3FlatBits);           // ... and start in front of the target
                    over = over || (k != 0 && pos + kS3FlatBits * k > target);
                    chain += 258u * k;
                    advance_flat(k);
                    pos = position();
                    go = go && pos < target;
                    rep += 5;
                }
            }
            c += chain << 6;
            bl += chain >= kS2LongRun ? chain / 16 - 1 : 0u;
            // on to the next chain only where the lane sits on a run token again (anything else: back to the look-ups);
            // the interval that starts behind this chain is then a chain and nothing else, in front of it zeros: marked.
            // (Only a chain that was cut off -- `go` still set: its last token was a 258-byte one -- is looked behind: a
            //  run token behind any other chain, which the reference's encoder never writes, parks the lane at its next
            //  look-up and is taken then.)
            bool again = false;
            if (__any(go)) {
                const S3Tok t = s3_token(canon, window30());
                again = go && more && chain != 0 && !stopped && !fault && pos < target && t.run != 0 && !t.bad;
            }
            cut(more && chain != 0, again ? pure_flag : 0u);
            more = again;
            if (__any(more)) refill();
        }
    };
    for (int round = 0; round < 4 && __any(need); round++) {
        {
            const uint32_t nxt = __shfl_down(start, 1, kWave);
            target = lane == last ? in_bits : nxt;
        }
        if (need) {
            slot = kS2HeadSlots;
            m = 0;
            bl = 0;
            c &= 63u;
            over = false;
            if (round > 0) {  // what the chain from the wrong start ran into (a stray end-of-block code, a bad token) is void
                stopped = false;
                fault = false;
                eob_bits = 0;
            }
        }
        cut(need);
        uint32_t dbg_periods = 0, dbg_special = 0;
        (void)dbg_periods;
        (void)dbg_special;
        // whole periods
        for (;;) {
            refill();
            const bool bulk = need && !stopped && !fault && pos + kS3PeriodBits <= target;
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
        // half a period, a quarter, ... a pair: a lane takes each size at most once on its way to the target (and again
        // behind a run chain), so the lanes are through in a handful of trips (rounds 5: groups of 8, then pairs: ~11)
    #pragma unroll 1
        for (uint32_t pairs = kS3PeriodPairs / 2; pairs != 0; pairs >>= 1) {
            const uint32_t bits = 2 * pairs * kLitBits;
            for (;;) {
                const bool act = need && !stopped && !fault && pos + bits <= target;
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
            const bool act = need && !stopped && !fault && pos < target;
            if (!__any(act)) break;
            const uint32_t w = window30();
            const uint32_t e = lit[w & (kLitSize - 1)];
            const bool spec = act && e == 0;
            const bool step = act && e != 0;
            cut(step && m + 1 > kS2Meter);
            m += step ? 1u : 0u;
            const uint32_t used = e & 15u, d = target - pos;
            const uint32_t len1 = canon[w & (kLitSize - 1)] >> 24;
            const bool one = step && used > d && len1 <= d;
            const bool full = step && !one;  // (also the step that goes over: no token ends on the target)
            over = over || (full && used > d);
            c += (full ? (e >> 6) & 3u : (one ? 1u : 0u)) << 6;
            advance(full ? used : (one ? len1 : 0u));
            pos = position();
            if (__any(spec)) special(spec);
        }
        if (need) landed = !fault && (lane == last ? stopped : (!stopped && (pos == target || over)));
        cut(need && (m > 0 || slot == kS2HeadSlots + 1));  // the end of the chain is a checkpoint too (unless a cut just made it one)
        // a chain that went over its target: the lane to its right counts again, from where that chain ended
        const uint32_t lpos = __shfl_up(pos, 1, kWave);
        const bool lover = __shfl_up((uint32_t)(need && over && !fault), 1, kWave) != 0;
        need = in_range && lane > 0 && lover;
        S3STAT(3, __popcll(__ballot(need)));
        if (__any(need)) {
            if (need) {
                start = lpos;
                landed = false;
            }
            start_reader(need, lpos);
            pos = position();
        }
    }
    S3T(6);
    // ---- the plan ----
    bool ok = !__any(!landed) && !__any(fault) && !__any(need);
    const uint32_t count = in_range ? c >> 6 : 0u;
    const uint32_t n_int = in_range ? slot - kS2HeadSlots - 1 : 0u;
    ok = ok && !__any(in_range && slot < kS2HeadSlots + 2);
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
    const uint32_t eob_end = __builtin_amdgcn_readlane(pos + eob_bits, last);
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

// Lean writing pass: the rounds of seg2_write for a stream whose run chains all stay inside the image (no bulk
// lines, no breaks) and whose slot is 16-B aligned -- every PNG-filter stream of the ultra-fast encoder with noisy
// rows.  A round is an order of magnitude fewer instructions around the look-up group than the general writer's:
//   * a run chain is decoded from the reference-layout table in the LDS (s3_token) and, being a run of zeros --
//     all the reference's encoder ever emits (src/compress/ultrafast.rs:46-66) --, leaves the zero-initialised
//     image as it is; a run of anything else sends the stream to the general writers behind this kernel;
//   * image positions are output positions, the Adler-32 terms are 32-bit sums and one 64-bit multiply-add per piece;
//   * the next round's input image is requested as soon as this round has decoded (the request is inline asm: the
//     compiler does not wait for it in front of the LDS reads of the flush), the flush comes behind it.
// False: the stream was left PENDING.
__device__ __forceinline__ bool seg3_write(const SegArgs& a, const uint32_t* lit, const uint32_t* canon, uint32_t* imgA, uint32_t* imgB,
                                           const uint2* ckpt, const uint64_t sid, const S2Plan& plan) {
    const uint32_t total = uni(plan.total), ni = uni(plan.ni), tb = uni(plan.tb), seg = uni(plan.seg);
    const int lane = threadIdx.x & (kWave - 1);
    const uint8_t* in = a.in + a.in_off[sid];
    uint8_t* const op = a.out + a.out_off[sid];
    const uint32_t ldsA = lds_offset(imgA), ldsB = lds_offset(imgB);
    uint8_t* const imgB8 = reinterpret_cast<uint8_t*>(imgB);
    uint32_t f0 = 0, qa = 0;
    uint32_t ad_a = 0, ad_u = 0;
    unsigned long long ad_b = 0;
    bool bad = false;
    for (uint32_t x = 16 * (uint32_t)lane; x < kS2OutCap; x += 16 * kWave)
        *reinterpret_cast<uint4*>(imgB8 + x) = make_uint4(0, 0, 0, 0);

    // The intervals of a round: interval f belongs to the last lane whose P <= f; its two checkpoints are requested,
    // what turns them into stream bits / output bytes is kept.  The owners of 64 consecutive intervals are a handful
    // of consecutive lanes, starting at the owner of the round's first interval -- known from the round before: the
    // P of the next eight lanes is read into scalar registers and compared (no dependent trips to the LDS crossbar;
    // rounds 3-4: a six-step binary search by ds_bpermute); only when a ninth owner could be among them (lanes with
    // a single interval: tiny streams) the search runs.
    struct Fetch {
        uint2 e0, e1;
        uint32_t pbase, qbase, sg;
        bool valid;
    };
    auto fetch = [&](uint32_t fbase, uint32_t sg_first) __attribute__((always_inline)) {
        Fetch t;
        const uint32_t f = fbase + (uint32_t)lane;
        t.valid = f < ni;
        uint32_t sg = sg_first;  // (uniform) owner of interval fbase, or a lane in front of it
        bool beyond = false;
#pragma unroll
        for (uint32_t j = 1; j <= 9; j++) {
            const uint32_t probe = sg_first + j;
            const uint32_t pv = probe < (uint32_t)kWave ? (uint32_t)__builtin_amdgcn_readlane((int)plan.P, (int)(probe & 63)) : 0xFFFFFFFFu;
            if (j <= 8) sg += pv <= f ? 1u : 0u;
            else beyond = pv <= f;
        }
        if (__any(beyond && t.valid)) {
            sg = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint32_t probe = sg + step;
                const uint32_t pv = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((probe & 63) << 2), (int)plan.P);
                if (probe < (uint32_t)kWave && pv <= f) sg = probe;
            }
        }
        t.sg = sg;
        const uint32_t sP = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(sg << 2), (int)plan.P);
        t.qbase = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(sg << 2), (int)plan.obase);
        t.pbase = a.canon_bits + sg * seg;
        const uint32_t slot = kS2HeadSlots + (f - sP);
        t.e0 = t.e1 = make_uint2(0, 0);
        if (t.valid) {
            t.e0 = ckpt[S2_CK_AT(sg, slot)];
            t.e1 = ckpt[S2_CK_AT(sg, slot + 1)];
        }
        return t;
    };
    // the intervals of the round in c0 (start) / c1 (end): x = stream bit of the first token, y = output bytes in
    // front of it; how many of them fit the two images: n
    // (a checkpoint holds its bytes less the whole lines its run chains left out of the general writer's image: here
    //  image positions are output positions, the lines are added back)
    uint2 c0, c1;
    bool valid, pure;
    auto settle = [&](const Fetch& t) __attribute__((always_inline)) {
        const uint32_t posmask = (1u << kS2PosBits) - 1;
        c0 = make_uint2(t.pbase + (t.e0.x & posmask), t.qbase + (t.e0.y & ~kS3PureFlag) + 16 * (t.e0.x >> kS2PosBits));
        c1 = make_uint2(t.pbase + (t.e1.x & posmask), t.qbase + (t.e1.y & ~kS3PureFlag) + 16 * (t.e1.x >> kS2PosBits));
        valid = t.valid;
        pure = t.valid && (t.e0.y & kS3PureFlag) != 0;
    };
    uint32_t own = 0;  // per lane: the owner of this lane's interval of the current round
    {
        const Fetch t0 = fetch(0, 0);
        settle(t0);
        own = t0.sg;
    }
    uint32_t n = 0, ib = 0;
    const uint8_t* a0 = in;
#ifdef FDH_S3_DEBUG
    uint32_t s3_why[3] = {0, 0, 0};
#endif
    auto stage_c = [&](uint32_t wq_) __attribute__((always_inline)) {
        const uint8_t* g0 = in + ((c0.x - 2) >> 3);
        a0 = reinterpret_cast<const uint8_t*>(uni64(reinterpret_cast<uintptr_t>(g0)) & ~(uintptr_t)15);
        ib = (uint32_t)(g0 - a0);
        // (what a lane writes into the image are the literals of its one group, at most 3 kS2Meter bytes from its start:
        //  a run chain behind them leaves the image as it is, however long, and a chain behind a chain needs no image)
        const bool fits = valid && ib + kS2InReach <= kS3InCap && (pure || (c0.y - wq_) + kS2OutReach <= kS2OutCap);
        const uint64_t fit_mask = __ballot(fits);
        n = fit_mask == ~0ull ? (uint32_t)kWave : (uint32_t)__builtin_ctzll(~fit_mask);
#ifdef FDH_S3_DEBUG
        if (n < (uint32_t)kWave) {  // why the round ends in front of lane n: no interval left / the input image / the output image
            const bool v_ = __builtin_amdgcn_readlane((int)valid, (int)n) != 0;
            const bool i_ = __builtin_amdgcn_readlane((int)(ib + kS2InReach <= kS3InCap), (int)n) != 0;
            s3_why[!v_ ? 0 : (!i_ ? 1 : 2)]++;
        }
#endif
    };
    auto request = [&](const uint8_t* from) __attribute__((always_inline)) {
        const uint8_t* p = from + 16 * (uint32_t)lane;
        uint32_t m0_saved;
        asm volatile(
            "  s_mov_b32 %[sv], m0\n"
            "  s_mov_b32 m0, %[base]\n"
            "  s_nop 0\n"
            "  global_load_lds_dwordx4 %[p], off\n"
            "  global_load_lds_dwordx4 %[p], off offset:1024\n"
            "  global_load_lds_dwordx4 %[p], off offset:2048\n"
            "  s_mov_b32 m0, %[sv]\n"
            : [sv] "=&s"(m0_saved)
            : [p] "v"(p), [base] "s"(uni(ldsA))
            : "memory");
    };
    stage_c(qa - 16);
    wave_sync();  // (the counting pass is done with this LDS)
    request(a0);
    uint32_t stores_behind = 0;
    S3W_DECL;
    uint32_t dbg_rounds = 0;
    (void)dbg_rounds;
    while (f0 < ni) {
        dbg_rounds++;
        S3W(7);
        const uint32_t wq = qa - 16;  // (mod 2^32: the image starts one piece in front of what has not been flushed)
        if (n == 0) {
            bad = true;
            break;
        }
        const bool act = (uint32_t)lane < n;
        const uint32_t pos0 = c0.x, pos1 = c1.x, q0 = c0.y;
        const uint32_t in_off0 = (uint32_t)(a0 - in), ib_cur = ib;
        const uint32_t qf_new = __builtin_amdgcn_readlane(c1.y, (int)(n - 1));
        // ---- the next round's intervals are on their way while this one decodes ----
        const Fetch nx = fetch(f0 + n, (uint32_t)__builtin_amdgcn_readlane((int)own, (int)min(n, (uint32_t)kWave - 1)));
        S3W(0);
        // ---- the input image (requested a round ago; the flush's stores and the two loads above came later) ----
        s2_wait_vm(stores_behind + (f0 + n < ni ? 2u : 0u));
        wave_sync();
        uint32_t wi = act ? (ib_cur >> 2) + 2 : 2u;
        uint32_t lo = imgA[wi - 2], hi = imgA[wi - 1];
        uint32_t boff = 8 * (ib_cur & 3) + ((pos0 - 2) & 7);
        uint32_t oaddr = ldsB + (pure ? 16u : q0 - wq);  // (a chain behind a chain: its lane marks time, ORs nothing)
        S3W(1);
        {
            uint32_t c = boff | (oaddr << 6), acc = 0, ra = ldsA + 4 * wi;
            if (act) (void)seg2_write_group(kS2Meter / 2, lo, hi, c, ra, acc);
            wi = (ra - ldsA) >> 2;
            boff = c & 63u;
            oaddr = c >> 6;
        }
        // (the checkpoints asked for above have had the whole group to arrive; taking them HERE keeps the compiler from
        // waiting for them -- and with them for this round's stores and the next input image -- at the loop's head)
        S3W(2);
        asm volatile("" ::"v"(nx.e0.x), "v"(nx.e0.y), "v"(nx.e1.x), "v"(nx.e1.y));
        const uint32_t pos = 8 * (in_off0 + 4 * (wi - 2)) + boff + 2;  // stream bit of the lane's next token
        // ---- a lane that is not at the end of its interval sits on the run chain that ends it: the counting pass has
        //      decoded that chain from these very bits (pos1 is where it ends) and a run repeats the byte in front of it
        //      -- a zero, from any encoder that does what the reference's does (src/compress/ultrafast.rs:46-66), so the
        //      zero-initialised image is left as it is; a run of anything else sends the stream to the general writers ----
        {
            const bool mine = act && !pure && pos < pos1;
            wave_sync();
            const uint32_t xi = mine ? oaddr - ldsB : 16u;
            const uint32_t front = imgB8[xi - 1];
            bad = bad || (mine && (front != 0 || wq + xi == 0));
        }
        if (__any(bad)) {
            bad = true;
            break;
        }
        wave_sync();
        S3W(3);
        const bool final_round = f0 + n >= ni;
        const uint32_t qa_new = final_round ? (qf_new + 15) & ~15u : max(qa, qf_new & ~127u);
        const uint32_t xa_new = qa_new - wq, n_cur = n;
        // (a round that ends in a long run chain: the image ends in front of qa_new -- what lies beyond it are the
        //  chain's zeros, stored straight to the slot, 1 KiB per instruction, nothing for the Adler-32 sums; issued in
        //  front of the input request, so that the wait for that request need not count them)
        const uint32_t xa_img = min(xa_new, kS2OutCap);
        for (uint32_t x = xa_img + 16 * (uint32_t)lane; x < xa_new; x += 16 * kWave) {
            const uint32_t v = wq + x;
            if (v + 16 <= total) *reinterpret_cast<uint4*>(op + v) = make_uint4(0, 0, 0, 0);
        }
        // ---- the next round: its intervals, how many fit, its input bytes (this round is done with the image) ----
        settle(nx);
        own = nx.sg;
        if (!final_round) {
            stage_c(qa_new - 16);
            request(a0);
        }
        stores_behind = final_round ? 64u : (xa_img - 16 + 1023) / 1024;
        S3W(4);
        // ---- flush: whole 128-B lines of the image (everything once the stream ends), one store per KiB; two pieces
        //      per trip, both read before the first is used ----
        auto piece = [&](const uint4 q, const uint32_t v) __attribute__((always_inline)) {
            *reinterpret_cast<uint4*>(op + v) = q;
#ifdef FDH_X_NO_ADLER  // timing experiment (wrong checksum)
            return;
#endif
            uint32_t sum = bytesum4(q.x);
            sum = __builtin_amdgcn_sad_u8(q.y, 0u, sum);
            sum = __builtin_amdgcn_sad_u8(q.z, 0u, sum);
            sum = __builtin_amdgcn_sad_u8(q.w, 0u, sum);
            ad_u = bytedot4(q.x, 0x03020100u, ad_u);
            ad_u = bytedot4(q.y, 0x07060504u, ad_u);
            ad_u = bytedot4(q.z, 0x0b0a0908u, ad_u);
            ad_u = bytedot4(q.w, 0x0f0e0d0cu, ad_u);
            ad_a += sum;
            ad_b += (unsigned long long)(total - v) * sum;
        };
        for (uint32_t x = 16 + 16 * (uint32_t)lane; __any(x < xa_img); x += 32 * kWave) {
            const uint32_t x2 = x + 16 * kWave, v = wq + x, v2 = wq + x2;
            const bool g1 = x < xa_img && v + 16 <= total, g2 = x2 < xa_img && v2 + 16 <= total;
            uint4 q1 = make_uint4(0, 0, 0, 0), q2 = q1;
            if (g1) q1 = *reinterpret_cast<const uint4*>(imgB8 + x);
            if (g2) q2 = *reinterpret_cast<const uint4*>(imgB8 + x2);
            if (g1) piece(q1, v);
            if (__any(g2)) {
                if (g2) piece(q2, v2);
            }
        }
        S3W(5);
        if (final_round) {
            if (total & 15u) {  // the last piece of the stream: its own bytes only
                const uint32_t v = total & ~15u, x = v - wq;
                if ((uint32_t)lane == (((x - 16) >> 4) & 63u)) {
                    uint4 q = make_uint4(0, 0, 0, 0);  // (beyond the image: the zeros of the chain the stream ends with)
                    if (x + 16 <= kS2OutCap) q = *reinterpret_cast<const uint4*>(imgB8 + x);
                    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
                    for (uint32_t kk = 0; kk < (total & 15u); kk++) {
                        const uint32_t byte = (w[kk >> 2] >> (8 * (kk & 3))) & 0xFFu;
                        op[v + kk] = (uint8_t)byte;
                        ad_a += byte;
                        ad_b += (unsigned long long)(total - v - kk) * byte;
                    }
                }
            }
        } else {
            // ---- carry: the pieces from one below qa_new on move to the front of the image, the rest is zeroed ----
            const uint32_t src0 = xa_new - 16;
            const uint32_t keep_end = ((qf_new + 15) & ~15u) - wq;
            const uint32_t used_end = min(kS2OutCap, (keep_end + kS2OutReach + 15) & ~15u);
            const uint32_t sx = src0 + 16 * (uint32_t)lane;
            uint4 keep = make_uint4(0, 0, 0, 0);
            if (sx < keep_end && sx + 16 <= kS2OutCap) keep = *reinterpret_cast<const uint4*>(imgB8 + sx);
            wave_sync();  // every lane has read before any lane writes
            *reinterpret_cast<uint4*>(imgB8 + 16 * (uint32_t)lane) = keep;
            for (uint32_t x = 16 * (uint32_t)(lane + kWave); x < used_end; x += 16 * kWave)
                *reinterpret_cast<uint4*>(imgB8 + x) = make_uint4(0, 0, 0, 0);
            wave_sync();
        }
        qa = qa_new;
        f0 += n_cur;
        S3W(6);
    }
    S3W_OUT;
    S3N(12, dbg_rounds);
    S3N(13, ni);
#ifdef FDH_S3_DEBUG
    S3N(14, s3_why[1]);
    S3N(15, s3_why[2]);
#endif
    if (__any(bad)) {
        if (lane == 0) seg_leave_pending(a, sid);
        return false;
    }
    // ---- Adler-32: A = 1 + sum of bytes ; B = total + sum of (total - offset) x byte ----
    uint32_t pa = ad_a % kAdlerMod;
    uint32_t pb = (uint32_t)((ad_b - ad_u) % kAdlerMod);
    pa = wave_sum_u32(pa);
    pb = wave_sum_u32(pb);
    const uint32_t A = (1u + pa) % kAdlerMod;
    const uint32_t B = (uint32_t)(((uint64_t)total + pb) % kAdlerMod);
    const uint32_t adler = (B << 16) | A;
    if (lane == 0) {
        // src/decompress.rs:306-326: byte boundary, then the big-endian Adler-32; Ok / WrongChecksum is the comparison
        // (every token was decoded and the trailer is there: see seg2_write)
        const uint32_t stored = ((uint32_t)in[tb] << 24) | ((uint32_t)in[tb + 1] << 16) | ((uint32_t)in[tb + 2] << 8) | (uint32_t)in[tb + 3];
        a.status[sid] = (stored == adler || (a.flags & 1u)) ? (uint32_t)ST_OK : (uint32_t)ST_WRONG_CHECKSUM;
        a.out_len[sid] = total;
        if (a.adler) a.adler[sid] = adler;
    }
    return true;
}

// False: the stream was passed on (too short / not canonical / near the end of the batch buffer, or
// after a counting pass that did not land).
__device__ __forceinline__ bool seg3_decode(const SegArgs& a, Seg3Lds& L, uint2* ckpt, const uint64_t sid) {
    const uint32_t wid = threadIdx.x / kWave;
    uint32_t* const W = L.w[wid];
    S2Plan plan;
    bool lean = false;
    const bool planned = seg3_plan(a, L.lit, L.canon, W, ckpt, sid, plan, lean);
    // the writing pass of the interval decoder: output image in the first 5 KiB, input image in the last 3
    if (planned && (a.flags & 0x40000u)) {  // debug (FDH_FLAG_LANDING_COUNT_ONLY): time the counting pass alone
        if ((threadIdx.x & (kWave - 1)) == 0) seg_leave_pending(a, sid);
        return true;
    }
    if (planned) {
        if (lean) (void)seg3_write(a, L.lit, L.canon, W + kS2BWords, W, ckpt, sid, plan);
        else seg2_write<kS3InCap>(a, L.lit, W + kS2BWords, W, ckpt, sid, plan);
    }
    S3T(8);
    return planned;
}

}  // namespace fdh
