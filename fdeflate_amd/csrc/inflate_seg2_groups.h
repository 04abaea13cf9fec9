// inflate_seg2_groups.h -- the hand-scheduled look-up groups of the interval decoder
// (inflate_seg2.h): table entry format, counting group, writing group.
//
// Table entry (built while the canonical table is staged into LDS): a step that starts with a
// literal retires up to three literals whose codes fit the 12 index bits together,
//   byte 0     [3:0] stream bits of the whole step (2..12), [5:4] = 0, [7:6] literals (1..3)
//   byte 1-3   the literal bytes in output order (unused ones zero)
// and a step whose first symbol is a run length / end-of-block / impossible code is ALL ZERO: the
// groups below apply an entry without looking at it, and a zero entry changes nothing (no bits, no
// bytes, OR of zero) -- a lane that meets one marks time until the group ends; the token itself
// is decoded by the general step from the canonical table (inflate_tables.h layout) behind the group.
//
// Both groups keep ONE counter register `c` per lane:
//   c[5:0]   bit offset of the next token inside the 64-bit window {lo, hi} (0..31 between pairs)
//   c[31:6]  bytes: the output bytes counted so far (counting group) / the LDS byte address of the
//            next output byte (writing group)
// so that applying an entry is a single SDWA add of its byte 0 (bits into [5:0] -- at most 24 per
// pair of steps on top of <= 31, never carrying into bit 6 -- and literals into [7:6] upwards), the
// 64-bit shift of the window takes its amount straight from c (the instruction reads c[5:0]) and
// the writing group derives the accumulator shift and the output dword address from c as well.
// The window sits TWO BITS IN FRONT of the next token, so bits [13:2] of the shifted window are the
// 12 index bits and `shifted & 0x3ffc` is the byte offset of the entry (table at LDS offset 0).
//
// gfx950: a VALU write of VCC must be two instructions away from a VALU read of it; plain VOP2
// costs ~2 issue cycles per wavefront, VOP3 / SDWA / compares ~4 (tools/ubench).
#pragma once
#include "inflate_tables.h"

namespace fdh {

__device__ __forceinline__ uint32_t seg2_entry_build(const uint32_t* canon, uint32_t i) {
    uint32_t used = 0, n = 0, lits = 0;
    for (int k = 0; k < 3; k++) {
        // the next symbol at the zero-extended rest of the index: right whenever its code ends
        // inside the index bits (prefix code), which is what `used + len <= kLitBits` checks
        const uint32_t e = canon[i >> used];
        const uint32_t kind = (e >> 4) & 15;
        if (kind != K_LIT1 && kind != K_LIT2) break;
        const uint32_t len = (e >> 24) & 15;  // bits of the first symbol (both kinds, inflate_tables.h)
        if (used + len > (uint32_t)kLitBits) break;
        lits |= ((e >> 8) & 0xFF) << (8 * k);
        used += len;
        n++;
    }
    return n ? (used | (n << 6) | (lits << 8)) : 0u;
}

// Fixed VGPRs of the groups: aligned pairs for the window, the shifted window, {literals, 0} and
// the shifted literals.
#ifndef FDH_S2_R0
#define FDH_S2_R0 "120"
#define FDH_S2_R1 "121"
#define FDH_S2_R2 "122"
#define FDH_S2_R3 "123"
#define FDH_S2_R4 "124"
#define FDH_S2_R5 "125"
#define FDH_S2_R6 "126"
#define FDH_S2_R7 "127"
#endif
#define S2_WLO "v" FDH_S2_R0
#define S2_WHI "v" FDH_S2_R1
#define S2_WIN "v[" FDH_S2_R0 ":" FDH_S2_R1 "]"
#define S2_SHF "v[" FDH_S2_R2 ":" FDH_S2_R3 "]"
#define S2_SH0 "v" FDH_S2_R2
#define S2_VLO "v" FDH_S2_R4
#define S2_VHI "v" FDH_S2_R5
#define S2_V64 "v[" FDH_S2_R4 ":" FDH_S2_R5 "]"
#define S2_TLO "v" FDH_S2_R6
#define S2_THI "v" FDH_S2_R7
#define S2_T64 "v[" FDH_S2_R6 ":" FDH_S2_R7 "]"
#define S2_CLOBBER4 "v" FDH_S2_R0, "v" FDH_S2_R1, "v" FDH_S2_R2, "v" FDH_S2_R3
#define S2_CLOBBER8 S2_CLOBBER4, "v" FDH_S2_R4, "v" FDH_S2_R5, "v" FDH_S2_R6, "v" FDH_S2_R7

#define S2_ADD_BYTE0(C, E) "  v_add_u32_sdwa " C ", " C ", " E " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"

// ---- counting group: `pairs` pairs of look-ups, input from the lane's ring ([word][lane]
// layout, 16 words: word w of a lane at rb + (w & 15) * 256).  lo / hi: the window; ra: LDS address
// of the ring word that follows hi.  Every lane the caller enabled must have its input in the ring
// (3 dwords per pair of steps + 1).
// Returns the last entry looked up (0: the lane sits on a token that is not a literal).
__device__ __forceinline__ uint32_t seg2_count_group(uint32_t pairs, uint32_t rb, uint32_t& lo, uint32_t& hi, uint32_t& c,
                                                     uint32_t& ra) {
    uint32_t e, t, nw;
    const uint32_t k256 = 256u, mf00 = 0xf00u;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"  // nothing of the compiler's may be in flight: the waits below assume it
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
        // top the window up: bit 5 of c set -> {lo, hi} = {hi, next word}, one more word from the ring
        "  v_and_b32 %[t], 32, %[c]\n"
        "  v_cmp_ne_u32 vcc, 0, %[t]\n"
        "  v_and_b32 %[c], 0xffffffdf, %[c]\n"
        "  s_sub_u32 %[pairs], %[pairs], 1\n"
        "  v_cndmask_b32 " S2_WLO ", " S2_WLO ", " S2_WHI ", vcc\n"
        "  v_cndmask_b32 " S2_WHI ", " S2_WHI ", %[nw], vcc\n"
        "  v_cndmask_b32 %[t], 0, %[k256], vcc\n"
        "  v_add_u32 %[t], %[ra], %[t]\n"
        "  v_and_or_b32 %[ra], %[t], %[mf00], %[rb]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "  s_cmp_lg_u32 %[pairs], 0\n"
        "  s_cbranch_scc1 Lpair_%=\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 %[lo], " S2_WLO "\n"
        "  v_mov_b32 %[hi], " S2_WHI "\n"
        : [pairs] "+s"(pairs), [lo] "+v"(lo), [hi] "+v"(hi), [c] "+v"(c), [ra] "+v"(ra), [e] "=&v"(e), [t] "=&v"(t),
          [nw] "=&v"(nw)
        : [rb] "v"(rb), [k256] "v"(k256), [mf00] "s"(mf00)
        : "vcc", "scc", "memory", S2_CLOBBER4);
    return e;
}

// ---- writing group: `pairs` pairs of look-ups; input from a flat LDS image of the stream bytes
// (ra: LDS address of the dword that follows hi), output OR-ed into a zero-initialised LDS image of
// the output bytes: c[31:6] is the LDS address of the next output byte, `acc` holds the bytes of
// the dword that address lies in (zero elsewhere) and is OR-ed to that dword in every step; the
// bytes that did not fit become the next accumulator once the dword is complete.  OR makes every
// store idempotent: the partial accumulator may be stored any number of times, two lanes may share
// a dword, and a lane may run on past the end of its interval -- what it then decodes are the
// stream's real next symbols, i.e. exactly what the lane behind it stores to the same bytes.
// On return the accumulator has been stored; returns the last entry looked up.
#ifdef FDH_X_NO_OR  // timing experiment (wrong bytes): the writing group without its LDS atomics
#define S2_OR_STEP "  s_nop 0\n"
#define S2_WAIT_E "  s_waitcnt lgkmcnt(0)\n"
#else
#define S2_OR_STEP "  ds_or_b32 %[wa], %[x]\n"
#define S2_WAIT_E "  s_waitcnt lgkmcnt(1)\n"
#endif
__device__ __forceinline__ uint32_t seg2_write_group(uint32_t pairs, uint32_t& lo, uint32_t& hi, uint32_t& c, uint32_t& ra,
                                                     uint32_t& acc) {
    // The store of a step is issued BEHIND the table read of the next step (the LDS works in order:
    // an atomic in front of the read would delay what the whole chain waits for), and the wait is for
    // "all but the newest LDS operation".  x: the accumulator as it is stored; acc: as it goes on.
    uint32_t e, t, u, nw, wa, sa, sb, x = acc;
    const uint32_t k4 = 4u;
    asm volatile(
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 " S2_WLO ", %[lo]\n"
        "  v_mov_b32 " S2_WHI ", %[hi]\n"
        "  v_mov_b32 " S2_VHI ", 0\n"
        "  v_lshrrev_b32 %[u], 3, %[c]\n"
        "  v_and_b32 %[sa], 24, %[u]\n"
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_b32 %[wa], -4, %[u]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "Lpair_%=:\n"
        // step A: accumulator shift sa -> sb
        "  v_lshrrev_b64 " S2_SHF ", %[c], " S2_WIN "\n"
        "  v_and_b32 %[t], 0x3ffc, " S2_SH0 "\n"
        "  ds_read_b32 %[e], %[t]\n"
        S2_OR_STEP  // the step before (at the start: the accumulator as it came in)
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_b32 %[wa], -4, %[u]\n"
        S2_WAIT_E
        "  v_lshrrev_b32 " S2_VLO ", 8, %[e]\n"
        "  v_lshlrev_b64 " S2_T64 ", %[sa], " S2_V64 "\n"
        "  v_or_b32 %[x], %[acc], " S2_TLO "\n"
        S2_ADD_BYTE0("%[c]", "%[e]")
        "  v_lshrrev_b32 %[u], 3, %[c]\n"
        "  v_and_b32 %[sb], 24, %[u]\n"
        "  v_cmp_lt_u32 vcc, %[sb], %[sa]\n"  // the shift wrapped: the dword is complete
        // step B: sb -> sa
        "  v_lshrrev_b64 " S2_SHF ", %[c], " S2_WIN "\n"
        "  v_and_b32 %[t], 0x3ffc, " S2_SH0 "\n"
        "  v_cndmask_b32 %[acc], %[x], " S2_THI ", vcc\n"
        "  ds_read_b32 %[e], %[t]\n"
        S2_OR_STEP
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_b32 %[wa], -4, %[u]\n"
        S2_WAIT_E
        "  v_lshrrev_b32 " S2_VLO ", 8, %[e]\n"
        "  v_lshlrev_b64 " S2_T64 ", %[sb], " S2_V64 "\n"
        "  v_or_b32 %[x], %[acc], " S2_TLO "\n"
        S2_ADD_BYTE0("%[c]", "%[e]")
        "  v_lshrrev_b32 %[u], 3, %[c]\n"
        "  v_and_b32 %[sa], 24, %[u]\n"
        "  v_cmp_lt_u32 vcc, %[sa], %[sb]\n"
        // top the window up
        "  v_and_b32 %[t], 32, %[c]\n"
        "  v_and_b32 %[c], 0xffffffdf, %[c]\n"
        "  v_cndmask_b32 %[acc], %[x], " S2_THI ", vcc\n"
        "  v_cmp_ne_u32 vcc, 0, %[t]\n"
        "  s_sub_u32 %[pairs], %[pairs], 1\n"
        "  s_cmp_lg_u32 %[pairs], 0\n"
        "  v_cndmask_b32 " S2_WLO ", " S2_WLO ", " S2_WHI ", vcc\n"
        "  v_cndmask_b32 " S2_WHI ", " S2_WHI ", %[nw], vcc\n"
        "  v_cndmask_b32 %[t], 0, %[k4], vcc\n"
        "  v_add_u32 %[ra], %[ra], %[t]\n"
        "  ds_read_b32 %[nw], %[ra]\n"
        "  s_cbranch_scc1 Lpair_%=\n"
        // the last step's store, and the bytes it left in the accumulator
        "  ds_or_b32 %[wa], %[x]\n"
        "  v_lshrrev_b32 %[u], 6, %[c]\n"
        "  v_and_b32 %[wa], -4, %[u]\n"
        "  ds_or_b32 %[wa], %[acc]\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_mov_b32 %[lo], " S2_WLO "\n"
        "  v_mov_b32 %[hi], " S2_WHI "\n"
        : [pairs] "+s"(pairs), [lo] "+v"(lo), [hi] "+v"(hi), [c] "+v"(c), [ra] "+v"(ra), [acc] "+v"(acc), [x] "+v"(x),
          [e] "=&v"(e), [t] "=&v"(t), [u] "=&v"(u), [nw] "=&v"(nw), [wa] "=&v"(wa), [sa] "=&v"(sa), [sb] "=&v"(sb)
        : [k4] "v"(k4)
        : "vcc", "scc", "memory", S2_CLOBBER8);
    return e;
}

}  // namespace fdh
