/*
 * fdeflate_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See fdeflate_oracle.h for scope, the pinning vectors and the usage rules.
 *
 * Plain-C restatement of image-rs/fdeflate 0.4.0-dev (reference at /root/reference):
 *   src/tables.rs, src/lib.rs:103-127, src/huffman.rs, src/decompress.rs,
 *   src/compress/ultrafast.rs, src/compress/mod.rs (level 0 / empty input only).
 */
#include "fdeflate_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* Entry flags: src/decompress.rs:61-63 */
#define LITERAL_ENTRY 0x8000u
#define EXCEPTIONAL_ENTRY 0x4000u
#define SECONDARY_TABLE_ENTRY 0x2000u

#define LITLEN_TABLE_SIZE 4096u /* src/decompress.rs:66 */
#define DIST_TABLE_SIZE 512u    /* src/decompress.rs:67 */
#define LITLEN_TABLE_BITS 12
#define DIST_TABLE_BITS 9

/* ------------------------------------------------------------------------- */
/* Constant tables: src/tables.rs (data)                                      */
/* ------------------------------------------------------------------------- */

/* src/tables.rs:7-20 */
static const uint8_t HUFFMAN_LENGTHS[286] = {
    2, 3, 4, 5, 5, 6, 6, 7, 7, 7, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9,
    10, 10, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12,
    12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 11, 11, 11, 11, 11, 11,
    11, 11, 11, 11, 10, 11, 10, 10, 10, 10, 10, 10, 10, 10, 10, 9, 9, 9, 9, 9, 8, 9,
    8, 8, 8, 8, 8, 7, 7, 7, 6, 6, 6, 5, 4, 3, 12, 12, 12, 9, 9, 11, 10, 11,
    11, 10, 11, 11, 11, 11, 11, 11, 12, 11, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 12, 9,
};

/* src/tables.rs:63-65 */
static const uint8_t CLCL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* src/tables.rs:68-70 */
static const uint8_t LEN_SYM_TO_LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                                  2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
/* src/tables.rs:73-76 */
static const uint16_t LEN_SYM_TO_LEN_BASE[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,
                                                  15, 17, 19, 23, 27, 31, 35, 43, 51,  59,
                                                  67, 83, 99, 115, 131, 163, 195, 227, 258};
/* src/tables.rs:79-82 */
static const uint8_t DIST_SYM_TO_DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                                    6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
/* src/tables.rs:85-88 */
static const uint16_t DIST_SYM_TO_DIST_BASE[30] = {
    1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
    193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};

/* src/compress/ultrafast.rs:82-86 */
static const uint8_t ULTRAFAST_HEADER[54] = {
    120, 1,   237, 192, 3,   160, 36,  89,  150, 198, 241, 255, 119, 238, 141, 200, 204, 167,
    114, 75,  99,  174, 109, 219, 182, 109, 219, 182, 109, 219, 182, 109, 105, 140, 158, 150,
    74,  175, 158, 50,  51,  34,  238, 249, 118, 183, 106, 122, 166, 135, 59,  107, 213, 15,
};

static uint16_t HUFFMAN_CODES[286];          /* src/tables.rs:22-25 */
static uint16_t LENGTH_TO_SYMBOL[256];       /* src/tables.rs:28-43 (derived from RFC 1951) */
static uint8_t LENGTH_TO_LEN_EXTRA[256];     /* src/tables.rs:46-55 */
static uint32_t LITLEN_TABLE_ENTRIES[288];   /* src/tables.rs:99-122 */
static uint32_t DISTANCE_TABLE_ENTRIES[32];  /* src/tables.rs:130-140 */

static pthread_once_t g_tables_once = PTHREAD_ONCE_INIT;

static uint16_t reverse_bits16(uint16_t v) {
    uint16_t r = 0;
    for (int i = 0; i < 16; i++) {
        r = (uint16_t)((r << 1) | ((v >> i) & 1));
    }
    return r;
}

/* compute_codes: src/lib.rs:103-127 */
static int compute_codes(const uint8_t *lengths, size_t n, uint16_t *codes) {
    uint32_t code = 0;
    for (unsigned len = 1; len <= 16; len++) {
        for (size_t i = 0; i < n; i++) {
            if (lengths[i] == len) {
                codes[i] = (uint16_t)(reverse_bits16((uint16_t)code) >> (16 - len));
                code += 1;
            }
        }
        code <<= 1;
    }
    return code == (2u << 16);
}

static void init_tables(void) {
    if (!compute_codes(HUFFMAN_LENGTHS, 286, HUFFMAN_CODES)) {
        abort(); /* "HUFFMAN_LENGTHS is invalid" src/tables.rs:24 */
    }
    /* LENGTH_TO_SYMBOL / LENGTH_TO_LEN_EXTRA, index = length - 3.  The literal arrays at
     * src/tables.rs:28-55 are the RFC-1951 length code map; index 255 (length 258) is
     * symbol 285 with 0 extra bits.  The reference's `tables` test
     * (src/decompress.rs:1198-1216) states exactly this derivation. */
    for (int sym = 0; sym < 29; sym++) {
        unsigned base = LEN_SYM_TO_LEN_BASE[sym];
        unsigned extra = LEN_SYM_TO_LEN_EXTRA[sym];
        for (unsigned j = 0; j < (1u << extra); j++) {
            if (sym == 27 && j == 31) {
                continue; /* length 258 belongs to symbol 285 */
            }
            LENGTH_TO_SYMBOL[base + j - 3] = (uint16_t)(257 + sym);
            LENGTH_TO_LEN_EXTRA[base + j - 3] = (uint8_t)extra;
        }
    }
    /* LITLEN_TABLE_ENTRIES: src/tables.rs:99-122 */
    for (int i = 0; i < 288; i++) {
        LITLEN_TABLE_ENTRIES[i] = EXCEPTIONAL_ENTRY;
    }
    for (uint32_t i = 0; i < 256; i++) {
        LITLEN_TABLE_ENTRIES[i] = (i << 16) | LITERAL_ENTRY | (1u << 8);
    }
    for (int i = 257; i < 286; i++) {
        LITLEN_TABLE_ENTRIES[i] = ((uint32_t)LEN_SYM_TO_LEN_BASE[i - 257] << 16) |
                                  ((uint32_t)LEN_SYM_TO_LEN_EXTRA[i - 257] << 8);
    }
    /* DISTANCE_TABLE_ENTRIES: src/tables.rs:130-140 */
    for (int i = 0; i < 32; i++) {
        DISTANCE_TABLE_ENTRIES[i] = 0;
    }
    for (int i = 0; i < 30; i++) {
        DISTANCE_TABLE_ENTRIES[i] = ((uint32_t)DIST_SYM_TO_DIST_BASE[i] << 16) |
                                    ((uint32_t)DIST_SYM_TO_DIST_EXTRA[i] << 8) | LITERAL_ENTRY;
    }
}

static void ensure_tables(void) { pthread_once(&g_tables_once, init_tables); }

const uint8_t *fdo_huffman_lengths(void) { return HUFFMAN_LENGTHS; }
const uint16_t *fdo_huffman_codes(void) {
    ensure_tables();
    return HUFFMAN_CODES;
}
const uint32_t *fdo_litlen_table_entries(void) {
    ensure_tables();
    return LITLEN_TABLE_ENTRIES;
}
const uint32_t *fdo_distance_table_entries(void) {
    ensure_tables();
    return DISTANCE_TABLE_ENTRIES;
}
const uint8_t *fdo_ultrafast_header(void) { return ULTRAFAST_HEADER; }

/* ------------------------------------------------------------------------- */
/* Adler-32 (RFC 1950; simd-adler32 produces the same function)               */
/* ------------------------------------------------------------------------- */

uint32_t fdo_adler32_update(uint32_t adler, const uint8_t *data, size_t len) {
    uint32_t a = adler & 0xffff, b = adler >> 16;
    while (len > 0) {
        size_t n = len < 5552 ? len : 5552; /* largest n with no u32 overflow */
        for (size_t i = 0; i < n; i++) {
            a += data[i];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        data += n;
        len -= n;
    }
    return (b << 16) | a;
}

uint32_t fdo_adler32(const uint8_t *data, size_t len) { return fdo_adler32_update(1, data, len); }

/* ------------------------------------------------------------------------- */
/* huffman.rs                                                                 */
/* ------------------------------------------------------------------------- */

/* next_codeword: src/huffman.rs:5-15 */
static uint16_t next_codeword(uint16_t codeword, uint16_t table_size) {
    if (codeword == (uint16_t)(table_size - 1)) {
        return codeword;
    }
    uint16_t x = (uint16_t)(codeword ^ (table_size - 1));
    /* adv = 15 - leading_zeros16(x) = index of highest set bit */
    int adv = 15;
    while (!((x >> adv) & 1)) {
        adv--;
    }
    uint16_t bit = (uint16_t)(1u << adv);
    codeword &= (uint16_t)(bit - 1);
    codeword |= bit;
    return codeword;
}

static uint32_t entry_for(const uint32_t *entries, size_t n_entries, size_t symbol) {
    /* entries.get(symbol).cloned().unwrap_or((symbol as u32) << 16) */
    return symbol < n_entries ? entries[symbol] : ((uint32_t)symbol << 16);
}

/* build_table: src/huffman.rs:18-184 */
int fdo_build_table(const uint8_t *lengths, size_t n_lengths, const uint32_t *entries,
                    size_t n_entries, uint16_t *codes, uint32_t *primary_table,
                    size_t primary_len, uint16_t *secondary, size_t secondary_cap,
                    size_t *secondary_len, int is_distance_table, int double_literal) {
    /* :27-31 */
    size_t histogram[16] = {0};
    for (size_t i = 0; i < n_lengths; i++) {
        histogram[lengths[i]] += 1;
    }
    /* :33-37 */
    size_t max_length = 15;
    while (max_length > 1 && histogram[max_length] == 0) {
        max_length -= 1;
    }
    /* :39-59 */
    if (is_distance_table) {
        if (max_length == 0) {
            memset(primary_table, 0, primary_len * sizeof(uint32_t));
            *secondary_len = 0;
            return 1;
        } else if (max_length == 1 && histogram[1] == 1) {
            size_t symbol = 0;
            while (lengths[symbol] != 1) {
                symbol++;
            }
            codes[symbol] = 0;
            uint32_t entry = entry_for(entries, n_entries, symbol) | 1u;
            for (size_t i = 0; i < primary_len; i += 2) {
                primary_table[i] = entry;
                if (i + 1 < primary_len) {
                    primary_table[i + 1] = 0;
                }
            }
            return 1; /* note: the secondary table is left untouched (:57) */
        }
    }
    /* :61-75 */
    size_t offsets[16] = {0};
    size_t codespace_used = 0;
    offsets[1] = histogram[0];
    for (size_t i = 1; i < max_length; i++) {
        offsets[i + 1] = offsets[i] + histogram[i];
        codespace_used = (codespace_used << 1) + histogram[i];
    }
    codespace_used = (codespace_used << 1) + histogram[max_length];
    if (codespace_used != ((size_t)1 << max_length)) {
        return 0;
    }
    /* :77-84 */
    size_t next_index[16];
    memcpy(next_index, offsets, sizeof(offsets));
    size_t sorted_symbols[288] = {0};
    for (size_t symbol = 0; symbol < n_lengths; symbol++) {
        uint8_t length = lengths[symbol];
        sorted_symbols[next_index[length]] = symbol;
        next_index[length] += 1;
    }
    /* :86-136 */
    uint16_t codeword = 0;
    size_t i = histogram[0];
    size_t primary_table_bits = 0;
    while (((size_t)1 << (primary_table_bits + 1)) <= primary_len) {
        primary_table_bits++;
    }
    size_t primary_table_mask = ((size_t)1 << primary_table_bits) - 1;
    for (size_t length = 1; length <= primary_table_bits; length++) {
        size_t current_table_end = (size_t)1 << length;
        for (size_t k = 0; k < histogram[length]; k++) {
            size_t symbol = sorted_symbols[i];
            i += 1;
            primary_table[codeword] = entry_for(entries, n_entries, symbol) | (uint32_t)length;
            codes[symbol] = codeword;
            codeword = next_codeword(codeword, (uint16_t)current_table_end);
        }
        if (double_literal) {
            for (size_t len1 = 1; len1 < length; len1++) {
                size_t len2 = length - len1;
                for (size_t s1i = offsets[len1]; s1i < next_index[len1]; s1i++) {
                    for (size_t s2i = offsets[len2]; s2i < next_index[len2]; s2i++) {
                        size_t sym1 = sorted_symbols[s1i];
                        size_t sym2 = sorted_symbols[s2i];
                        if (sym1 < 256 && sym2 < 256) {
                            uint16_t codeword1 = codes[sym1];
                            uint16_t codeword2 = codes[sym2];
                            uint16_t cw = (uint16_t)(codeword1 | (codeword2 << len1));
                            uint32_t entry = ((uint32_t)sym1 << 16) | ((uint32_t)sym2 << 24) |
                                             LITERAL_ENTRY | (2u << 8);
                            primary_table[cw] = entry | (uint32_t)length;
                        }
                    }
                }
            }
        }
        if (length < primary_table_bits) {
            memcpy(primary_table + current_table_end, primary_table,
                   current_table_end * sizeof(uint32_t));
        }
    }
    /* :138-181 */
    size_t sec_len = 0;
    if (max_length > primary_table_bits) {
        size_t subtable_start = 0;
        size_t subtable_prefix = (size_t)-1;
        for (size_t length = primary_table_bits + 1; length <= max_length; length++) {
            size_t subtable_size = (size_t)1 << (length - primary_table_bits);
            uint32_t overflow_bits_mask = (uint32_t)subtable_size - 1;
            for (size_t k = 0; k < histogram[length]; k++) {
                if ((codeword & primary_table_mask) != subtable_prefix) {
                    subtable_prefix = codeword & primary_table_mask;
                    subtable_start = sec_len;
                    primary_table[subtable_prefix] = ((uint32_t)subtable_start << 16) |
                                                     EXCEPTIONAL_ENTRY | SECONDARY_TABLE_ENTRY |
                                                     overflow_bits_mask;
                    if (subtable_start + subtable_size > secondary_cap) {
                        abort();
                    }
                    for (size_t z = sec_len; z < subtable_start + subtable_size; z++) {
                        secondary[z] = 0;
                    }
                    sec_len = subtable_start + subtable_size;
                }
                size_t symbol = sorted_symbols[i];
                i += 1;
                codes[symbol] = codeword;
                secondary[subtable_start + (codeword >> primary_table_bits)] =
                    (uint16_t)(((uint16_t)symbol << 4) | (uint16_t)length);
                codeword = next_codeword(codeword, (uint16_t)((size_t)1 << length));
            }
            if (length < max_length && (codeword & primary_table_mask) == subtable_prefix) {
                size_t cur = sec_len - subtable_start;
                if (sec_len + cur > secondary_cap) {
                    abort();
                }
                memcpy(secondary + sec_len, secondary + subtable_start, cur * sizeof(uint16_t));
                sec_len += cur;
                size_t new_size = sec_len - subtable_start;
                uint32_t mask2 = (uint32_t)new_size - 1;
                primary_table[subtable_prefix] = ((uint32_t)subtable_start << 16) |
                                                 EXCEPTIONAL_ENTRY | SECONDARY_TABLE_ENTRY | mask2;
            }
        }
    }
    *secondary_len = sec_len;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* decompress.rs                                                              */
/* ------------------------------------------------------------------------- */

#define SECONDARY_CAP 4096 /* generous: <= 288 long codes, subtables <= 8 entries each doubling */

typedef struct {
    uint64_t buffer;
    uint8_t nbits;
} BitBuffer; /* src/decompress.rs:1022-1025 */

typedef struct {
    const uint8_t *ptr;
    size_t len;
} Slice;

/* fill_buffer: src/decompress.rs:1035-1052 */
static void fill_buffer(BitBuffer *b, Slice *input) {
    if (input->len >= 8) {
        uint8_t bits = b->nbits & 63;
        uint64_t v;
        memcpy(&v, input->ptr, 8); /* little-endian host assumed (x86-64) */
        b->buffer |= v << bits;
        size_t adv = (size_t)((63 - bits) / 8);
        input->ptr += adv;
        input->len -= adv;
        bits |= 56;
        b->nbits = bits;
    } else {
        size_t room = (size_t)((63 - b->nbits) / 8);
        size_t nbytes = input->len < room ? input->len : room;
        uint8_t tmp[8] = {0};
        memcpy(tmp, input->ptr, nbytes);
        uint64_t v;
        memcpy(&v, tmp, 8);
        b->buffer |= (b->nbits < 64) ? (v << b->nbits) : 0; /* checked_shl(..).unwrap_or(0) */
        b->nbits = (uint8_t)(b->nbits + nbytes * 8);
        input->ptr += nbytes;
        input->len -= nbytes;
    }
}

/* peek_bits / consume_bits: src/decompress.rs:1054-1063 */
static uint64_t peek_bits(const BitBuffer *b, uint8_t nbits) {
    return b->buffer & (((uint64_t)1 << nbits) - 1);
}
static void consume_bits(BitBuffer *b, uint8_t nbits) {
    b->buffer >>= nbits;
    b->nbits = (uint8_t)(b->nbits - nbits);
}

enum { Q_NONE = 0, Q_RLE = 1, Q_BACKREF = 2 }; /* QueuedOutput: src/decompress.rs:1067-1070 */

typedef enum {
    ST_ZLIB_HEADER,
    ST_BLOCK_HEADER,
    ST_CODE_LENGTH_CODES,
    ST_CODE_LENGTHS,
    ST_COMPRESSED_DATA,
    ST_UNCOMPRESSED_DATA,
    ST_CHECKSUM,
    ST_DONE
} State; /* src/decompress.rs:84-93 */

struct fdo_decompressor {
    /* CompressedBlock: src/decompress.rs:71-81 */
    uint32_t litlen_table[LITLEN_TABLE_SIZE];
    uint16_t secondary_table[SECONDARY_CAP];
    size_t secondary_len;
    uint32_t dist_table[DIST_TABLE_SIZE];
    uint16_t dist_secondary_table[SECONDARY_CAP];
    size_t dist_secondary_len;
    uint16_t eof_code, eof_mask;
    uint8_t eof_bits;
    /* BlockHeader: src/decompress.rs:50-59 */
    size_t hlit, hdist, hclen, num_lengths_read;
    uint32_t cl_table[128];
    uint8_t code_lengths[320];
    /* Decompressor: src/decompress.rs:96-113 */
    uint16_t uncompressed_bytes_left;
    BitBuffer bits;
    int q_kind;
    uint8_t q_data;
    size_t q_dist, q_length;
    int last_block, fixed_table;
    State state;
    uint32_t checksum;
    int ignore_adler32;
};

fdo_decompressor *fdo_decompressor_new(void) { /* src/decompress.rs:123-151 */
    ensure_tables();
    fdo_decompressor *d = (fdo_decompressor *)calloc(1, sizeof(*d));
    d->state = ST_ZLIB_HEADER;
    d->checksum = 1;
    return d;
}
void fdo_decompressor_free(fdo_decompressor *d) { free(d); }
void fdo_decompressor_ignore_adler32(fdo_decompressor *d) { d->ignore_adler32 = 1; }
int fdo_decompressor_is_done(const fdo_decompressor *d) { return d->state == ST_DONE; }

/* CompressedBlock::build_tables: src/decompress.rs:561-606 */
static int build_tables(fdo_decompressor *d, size_t hlit, const uint8_t *code_lengths) {
    if (code_lengths[256] == 0) {
        return FDO_BAD_LITERAL_LENGTH_HUFFMAN_TREE;
    }
    uint16_t codes[288] = {0};
    d->secondary_len = 0;
    if (!fdo_build_table(code_lengths, hlit, LITLEN_TABLE_ENTRIES, 288, codes, d->litlen_table,
                         LITLEN_TABLE_SIZE, d->secondary_table, SECONDARY_CAP, &d->secondary_len,
                         0, 1)) {
        return FDO_BAD_CODE_LENGTH_HUFFMAN_TREE; /* sic: src/decompress.rs:579 */
    }
    d->eof_code = codes[256];
    d->eof_mask = (uint16_t)((1u << code_lengths[256]) - 1);
    d->eof_bits = code_lengths[256];

    const uint8_t *lengths = code_lengths + 288;
    int all_zero = 1;
    for (int i = 0; i < 32; i++) {
        if (lengths[i] != 0) {
            all_zero = 0;
        }
    }
    if (all_zero) {
        memset(d->dist_table, 0, sizeof(d->dist_table));
    } else {
        uint16_t dist_codes[32] = {0};
        if (!fdo_build_table(lengths, 32, DISTANCE_TABLE_ENTRIES, 32, dist_codes, d->dist_table,
                             DIST_TABLE_SIZE, d->dist_secondary_table, SECONDARY_CAP,
                             &d->dist_secondary_len, 1, 0)) {
            return FDO_BAD_DISTANCE_HUFFMAN_TREE;
        }
    }
    return FDO_OK;
}

int fdo_build_decode_tables(size_t hlit, const uint8_t code_lengths[320], uint32_t *litlen,
                            uint32_t *dist, uint16_t *eof_code, uint16_t *eof_mask,
                            uint8_t *eof_bits) {
    fdo_decompressor *d = fdo_decompressor_new();
    int st = build_tables(d, hlit, code_lengths);
    if (st == FDO_OK) {
        memcpy(litlen, d->litlen_table, sizeof(d->litlen_table));
        memcpy(dist, d->dist_table, sizeof(d->dist_table));
        *eof_code = d->eof_code;
        *eof_mask = d->eof_mask;
        *eof_bits = d->eof_bits;
    }
    fdo_decompressor_free(d);
    return st;
}

/* The match copy shared by the fast and careful loops:
 * src/decompress.rs:792-829 and :969-1006.  Returns 1 if the caller must `break`
 * (output filled with a queued remainder), 0 otherwise. */
static int copy_match(fdo_decompressor *d, uint8_t *output, size_t output_len,
                      size_t *output_index_p, size_t length, size_t dist) {
    size_t output_index = *output_index_p;
    size_t copy_length = length < output_len - output_index ? length : output_len - output_index;
    if (dist == 1) {
        uint8_t last = output[output_index - 1];
        memset(output + output_index, last, copy_length);
        if (length - copy_length != 0) {
            d->q_kind = Q_RLE;
            d->q_data = last;
            d->q_length = length - copy_length;
            *output_index_p = output_len;
            return 1;
        }
    } else if (output_index + length + 15 <= output_len) {
        size_t start = output_index - dist;
        memmove(output + output_index, output + start, 16); /* copy_within */
        if (length > 16 || dist < 16) {
            size_t step = dist < 16 ? dist : 16;
            for (size_t i = step; i < length; i += step) { /* step_by(..).skip(1) */
                memmove(output + output_index + i, output + start + i, 16);
            }
        }
    } else {
        if (dist < copy_length) {
            for (size_t i = 0; i < copy_length; i++) {
                output[output_index + i] = output[output_index + i - dist];
            }
        } else {
            memmove(output + output_index, output + output_index - dist, copy_length);
        }
        if (length - copy_length != 0) {
            d->q_kind = Q_BACKREF;
            d->q_dist = dist;
            d->q_length = length - copy_length;
            *output_index_p = output_len;
            return 1;
        }
    }
    *output_index_p = output_index + copy_length;
    return 0;
}

enum { MORE_DATA_PRESENT = 0, REACHED_END_OF_BLOCK = 1 };

/* CompressedBlock::read_compressed: src/decompress.rs:611-1018.
 * Returns a status code; *block_status and *output_index_p are the Ok tuple. */
static int read_compressed(fdo_decompressor *d, Slice *remaining_input, uint8_t *output,
                           size_t output_len, size_t *output_index_p, int *block_status) {
    BitBuffer *bb = &d->bits;
    const uint64_t litlen_table_mask = LITLEN_TABLE_SIZE - 1;
    const unsigned litlen_table_bits = LITLEN_TABLE_BITS;
    const uint64_t dist_table_mask = DIST_TABLE_SIZE - 1;
    const unsigned dist_table_bits = DIST_TABLE_BITS;
    size_t output_index = *output_index_p;

    /* Fast decoding loop: :645-830 */
    fill_buffer(bb, remaining_input);
    uint32_t litlen_entry = d->litlen_table[bb->buffer & litlen_table_mask];
    while (output_index + 8 <= output_len && remaining_input->len >= 8) {
        uint64_t bits;
        uint8_t litlen_code_bits = (uint8_t)litlen_entry;
        if (litlen_entry & LITERAL_ENTRY) {
            uint32_t litlen_entry2 =
                d->litlen_table[(bb->buffer >> litlen_code_bits) & litlen_table_mask];
            uint8_t litlen_code_bits2 = (uint8_t)litlen_entry2;
            uint32_t litlen_entry3 =
                d->litlen_table[(bb->buffer >> (uint8_t)(litlen_code_bits + litlen_code_bits2)) &
                                litlen_table_mask];
            uint8_t litlen_code_bits3 = (uint8_t)litlen_entry3;
            uint32_t litlen_entry4 =
                d->litlen_table[(bb->buffer >> (uint8_t)(litlen_code_bits + litlen_code_bits2 +
                                                         litlen_code_bits3)) &
                                litlen_table_mask];

            size_t advance_output_bytes = (litlen_entry & 0xf00) >> 8;
            output[output_index] = (uint8_t)(litlen_entry >> 16);
            output[output_index + 1] = (uint8_t)(litlen_entry >> 24);
            output_index += advance_output_bytes;

            if (litlen_entry2 & LITERAL_ENTRY) {
                size_t advance_output_bytes2 = (litlen_entry2 & 0xf00) >> 8;
                output[output_index] = (uint8_t)(litlen_entry2 >> 16);
                output[output_index + 1] = (uint8_t)(litlen_entry2 >> 24);
                output_index += advance_output_bytes2;

                if (litlen_entry3 & LITERAL_ENTRY) {
                    size_t advance_output_bytes3 = (litlen_entry3 & 0xf00) >> 8;
                    output[output_index] = (uint8_t)(litlen_entry3 >> 16);
                    output[output_index + 1] = (uint8_t)(litlen_entry3 >> 24);
                    output_index += advance_output_bytes3;

                    litlen_entry = litlen_entry4;
                    consume_bits(bb, (uint8_t)(litlen_code_bits + litlen_code_bits2 +
                                               litlen_code_bits3));
                    fill_buffer(bb, remaining_input);
                    continue;
                } else {
                    consume_bits(bb, (uint8_t)(litlen_code_bits + litlen_code_bits2));
                    litlen_entry = litlen_entry3;
                    litlen_code_bits = litlen_code_bits3;
                    fill_buffer(bb, remaining_input);
                    bits = bb->buffer;
                }
            } else {
                consume_bits(bb, litlen_code_bits);
                bits = bb->buffer;
                litlen_entry = litlen_entry2;
                litlen_code_bits = litlen_code_bits2;
                if (bb->nbits < 48) {
                    fill_buffer(bb, remaining_input);
                }
            }
        } else {
            bits = bb->buffer;
        }

        /* :708-748 */
        uint32_t length_base;
        uint8_t length_extra_bits;
        if ((litlen_entry & EXCEPTIONAL_ENTRY) == 0) {
            length_base = litlen_entry >> 16;
            length_extra_bits = (uint8_t)(litlen_entry >> 8);
        } else if (litlen_entry & SECONDARY_TABLE_ENTRY) {
            uint32_t secondary_table_index =
                (litlen_entry >> 16) + ((uint32_t)(bits >> litlen_table_bits) & (litlen_entry & 0xff));
            uint16_t secondary_entry = d->secondary_table[secondary_table_index];
            uint16_t litlen_symbol = secondary_entry >> 4;
            uint8_t code_bits = (uint8_t)(secondary_entry & 0xf);
            if (litlen_symbol <= 255) {
                consume_bits(bb, code_bits);
                litlen_entry = d->litlen_table[bb->buffer & litlen_table_mask];
                fill_buffer(bb, remaining_input);
                output[output_index] = (uint8_t)litlen_symbol;
                output_index += 1;
                continue;
            } else if (litlen_symbol == 256) {
                consume_bits(bb, code_bits);
                *output_index_p = output_index;
                *block_status = REACHED_END_OF_BLOCK;
                return FDO_OK;
            } else {
                length_base = LEN_SYM_TO_LEN_BASE[litlen_symbol - 257];
                length_extra_bits = LEN_SYM_TO_LEN_EXTRA[litlen_symbol - 257];
                litlen_code_bits = code_bits;
            }
        } else if (litlen_code_bits == 0) {
            return FDO_INVALID_LITERAL_LENGTH_CODE;
        } else {
            consume_bits(bb, litlen_code_bits);
            *output_index_p = output_index;
            *block_status = REACHED_END_OF_BLOCK;
            return FDO_OK;
        }
        bits >>= litlen_code_bits;

        uint64_t length_extra_mask = ((uint64_t)1 << length_extra_bits) - 1;
        size_t length = (size_t)length_base + (size_t)(bits & length_extra_mask);
        bits >>= length_extra_bits;

        /* :755-781 */
        uint32_t dist_entry = d->dist_table[bits & dist_table_mask];
        uint16_t dist_base;
        uint8_t dist_extra_bits, dist_code_bits;
        if (dist_entry & LITERAL_ENTRY) {
            dist_base = (uint16_t)(dist_entry >> 16);
            dist_extra_bits = (uint8_t)(dist_entry >> 8) & 0xf;
            dist_code_bits = (uint8_t)dist_entry;
        } else if ((dist_entry >> 8) == 0) {
            return FDO_INVALID_DISTANCE_CODE;
        } else {
            uint32_t secondary_table_index =
                (dist_entry >> 16) + ((uint32_t)(bits >> dist_table_bits) & (dist_entry & 0xff));
            uint16_t secondary_entry = d->dist_secondary_table[secondary_table_index];
            size_t dist_symbol = secondary_entry >> 4;
            if (dist_symbol >= 30) {
                return FDO_INVALID_DISTANCE_CODE;
            }
            dist_base = DIST_SYM_TO_DIST_BASE[dist_symbol];
            dist_extra_bits = DIST_SYM_TO_DIST_EXTRA[dist_symbol];
            dist_code_bits = (uint8_t)(secondary_entry & 0xf);
        }
        bits >>= dist_code_bits;

        size_t dist = (size_t)dist_base + (size_t)(bits & (((uint64_t)1 << dist_extra_bits) - 1));
        if (dist > output_index) {
            return FDO_DISTANCE_TOO_FAR_BACK;
        }

        consume_bits(bb, (uint8_t)(litlen_code_bits + length_extra_bits + dist_code_bits +
                                   dist_extra_bits));
        fill_buffer(bb, remaining_input);
        litlen_entry = d->litlen_table[bb->buffer & litlen_table_mask];

        if (copy_match(d, output, output_len, &output_index, length, dist)) {
            break;
        }
    }

    /* Careful decoding loop: :836-1007 */
    for (;;) {
        fill_buffer(bb, remaining_input);
        if (output_index == output_len) {
            break;
        }

        uint64_t bits = bb->buffer;
        uint32_t entry = d->litlen_table[bits & litlen_table_mask];
        uint8_t litlen_code_bits = (uint8_t)entry;

        if (entry & LITERAL_ENTRY) {
            size_t advance_output_bytes = (entry & 0xf00) >> 8;
            if (bb->nbits < litlen_code_bits) {
                break;
            } else if (output_index + 1 < output_len) {
                output[output_index] = (uint8_t)(entry >> 16);
                output[output_index + 1] = (uint8_t)(entry >> 24);
                output_index += advance_output_bytes;
                consume_bits(bb, litlen_code_bits);
                continue;
            } else if (output_index + advance_output_bytes == output_len) {
                output[output_index] = (uint8_t)(entry >> 16);
                output_index += 1;
                consume_bits(bb, litlen_code_bits);
                break;
            } else {
                output[output_index] = (uint8_t)(entry >> 16);
                d->q_kind = Q_RLE;
                d->q_data = (uint8_t)(entry >> 24);
                d->q_length = 1;
                output_index += 1;
                consume_bits(bb, litlen_code_bits);
                break;
            }
        }

        uint32_t length_base;
        uint8_t length_extra_bits;
        if ((entry & EXCEPTIONAL_ENTRY) == 0) {
            length_base = entry >> 16;
            length_extra_bits = (uint8_t)(entry >> 8);
        } else if (entry & SECONDARY_TABLE_ENTRY) {
            uint32_t secondary_table_index =
                (entry >> 16) + ((uint32_t)(bits >> litlen_table_bits) & (entry & 0xff));
            uint16_t secondary_entry = d->secondary_table[secondary_table_index];
            uint16_t litlen_symbol = secondary_entry >> 4;
            uint8_t code_bits = (uint8_t)(secondary_entry & 0xf);

            if (bb->nbits < code_bits) {
                break;
            } else if (litlen_symbol < 256) {
                consume_bits(bb, code_bits);
                output[output_index] = (uint8_t)litlen_symbol;
                output_index += 1;
                continue;
            } else if (litlen_symbol == 256) {
                consume_bits(bb, code_bits);
                *output_index_p = output_index;
                *block_status = REACHED_END_OF_BLOCK;
                return FDO_OK;
            }
            length_base = LEN_SYM_TO_LEN_BASE[litlen_symbol - 257];
            length_extra_bits = LEN_SYM_TO_LEN_EXTRA[litlen_symbol - 257];
            litlen_code_bits = code_bits;
        } else if (litlen_code_bits == 0) {
            return FDO_INVALID_LITERAL_LENGTH_CODE;
        } else {
            if (bb->nbits < litlen_code_bits) {
                break;
            }
            consume_bits(bb, litlen_code_bits);
            *output_index_p = output_index;
            *block_status = REACHED_END_OF_BLOCK;
            return FDO_OK;
        }
        bits >>= litlen_code_bits;

        uint64_t length_extra_mask = ((uint64_t)1 << length_extra_bits) - 1;
        size_t length = (size_t)length_base + (size_t)(bits & length_extra_mask);
        bits >>= length_extra_bits;

        uint32_t dist_entry = d->dist_table[bits & dist_table_mask];
        uint16_t dist_base;
        uint8_t dist_extra_bits, dist_code_bits;
        if (dist_entry & LITERAL_ENTRY) {
            dist_base = (uint16_t)(dist_entry >> 16);
            dist_extra_bits = (uint8_t)(dist_entry >> 8) & 0xf;
            dist_code_bits = (uint8_t)dist_entry;
        } else if (bb->nbits > (uint8_t)(litlen_code_bits + length_extra_bits + dist_table_bits)) {
            if ((dist_entry >> 8) == 0) {
                return FDO_INVALID_DISTANCE_CODE;
            }
            uint32_t secondary_table_index =
                (dist_entry >> 16) + ((uint32_t)(bits >> dist_table_bits) & (dist_entry & 0xff));
            uint16_t secondary_entry = d->dist_secondary_table[secondary_table_index];
            size_t dist_symbol = secondary_entry >> 4;
            if (dist_symbol >= 30) {
                return FDO_INVALID_DISTANCE_CODE;
            }
            dist_base = DIST_SYM_TO_DIST_BASE[dist_symbol];
            dist_extra_bits = DIST_SYM_TO_DIST_EXTRA[dist_symbol];
            dist_code_bits = (uint8_t)(secondary_entry & 0xf);
        } else {
            break;
        }
        bits >>= dist_code_bits;

        size_t dist = (size_t)dist_base + (size_t)(bits & (((uint64_t)1 << dist_extra_bits) - 1));
        uint8_t total_bits =
            (uint8_t)(litlen_code_bits + length_extra_bits + dist_code_bits + dist_extra_bits);

        if (bb->nbits < total_bits) {
            break;
        } else if (dist > output_index) {
            return FDO_DISTANCE_TOO_FAR_BACK;
        }

        consume_bits(bb, total_bits);

        if (copy_match(d, output, output_len, &output_index, length, dist)) {
            break;
        }
    }

    /* :1009-1017 */
    *output_index_p = output_index;
    if (d->q_kind == Q_NONE && bb->nbits >= 15 &&
        ((uint16_t)peek_bits(bb, 15) & d->eof_mask) == d->eof_code) {
        consume_bits(bb, d->eof_bits);
        *block_status = REACHED_END_OF_BLOCK;
        return FDO_OK;
    }
    *block_status = MORE_DATA_PRESENT;
    return FDO_OK;
}

/* read_block_header: src/decompress.rs:344-438 */
static int read_block_header(fdo_decompressor *d, Slice *remaining_input) {
    for (;;) { /* the single tail-recursive call at :393 is a loop here */
        fill_buffer(&d->bits, remaining_input);
        if (d->bits.nbits < 10) {
            return FDO_OK;
        }
        uint64_t start = peek_bits(&d->bits, 3);
        d->last_block = (start & 1) != 0;
        switch (start >> 1) {
        case 0: {
            uint8_t align_bits = (uint8_t)((d->bits.nbits - 3) % 8);
            uint8_t header_bits = (uint8_t)(3 + 32 + align_bits);
            if (d->bits.nbits < header_bits) {
                return FDO_OK;
            }
            uint16_t len =
                (uint16_t)(peek_bits(&d->bits, (uint8_t)(align_bits + 19)) >> (align_bits + 3));
            uint16_t nlen = (uint16_t)(peek_bits(&d->bits, header_bits) >> (align_bits + 19));
            if (nlen != (uint16_t)~len) {
                return FDO_INVALID_UNCOMPRESSED_BLOCK_LENGTH;
            }
            d->state = ST_UNCOMPRESSED_DATA;
            d->uncompressed_bytes_left = len;
            consume_bits(&d->bits, header_bits);
            return FDO_OK;
        }
        case 1: {
            consume_bits(&d->bits, 3);
            if (peek_bits(&d->bits, 7) == 0) {
                consume_bits(&d->bits, 7);
                if (d->last_block) {
                    d->state = ST_CHECKSUM;
                    return FDO_OK;
                }
                while (d->bits.nbits >= 10 && peek_bits(&d->bits, 10) == 2) {
                    consume_bits(&d->bits, 10);
                    fill_buffer(&d->bits, remaining_input);
                }
                continue; /* return self.read_block_header(remaining_input) */
            }
            if (!d->fixed_table) {
                d->fixed_table = 1;
                /* The reference copies FIXED_LITLEN_TABLE x8 and FIXED_DIST_TABLE x16
                 * (:400-405); those constants equal build_tables(288, FIXED_CODE_LENGTHS)
                 * (src/decompress.rs:1218-1233, checked in tests/test_oracle_golden.py). */
                uint8_t fixed[320];
                int i = 0;
                for (; i < 144; i++) fixed[i] = 8;
                for (; i < 256; i++) fixed[i] = 9;
                for (; i < 280; i++) fixed[i] = 7;
                for (; i < 288; i++) fixed[i] = 8;
                for (; i < 320; i++) fixed[i] = 5;
                if (build_tables(d, 288, fixed) != FDO_OK) {
                    abort();
                }
                d->eof_bits = 7;
                d->eof_code = 0;
                d->eof_mask = 0x7f;
            }
            d->state = ST_COMPRESSED_DATA;
            return FDO_OK;
        }
        case 2: {
            if (d->bits.nbits < 17) {
                return FDO_OK;
            }
            d->hlit = (size_t)(peek_bits(&d->bits, 8) >> 3) + 257;
            d->hdist = (size_t)(peek_bits(&d->bits, 13) >> 8) + 1;
            d->hclen = (size_t)(peek_bits(&d->bits, 17) >> 13) + 4;
            if (d->hlit > 286) {
                return FDO_INVALID_HLIT;
            }
            if (d->hdist > 30) {
                return FDO_INVALID_HDIST;
            }
            consume_bits(&d->bits, 17);
            d->state = ST_CODE_LENGTH_CODES;
            d->fixed_table = 0;
            return FDO_OK;
        }
        default:
            return FDO_INVALID_BLOCK_TYPE;
        }
    }
}

/* read_code_length_codes: src/decompress.rs:440-477 */
static int read_code_length_codes(fdo_decompressor *d, Slice *remaining_input) {
    fill_buffer(&d->bits, remaining_input);
    if ((size_t)d->bits.nbits + remaining_input->len * 8 < 3 * d->hclen) {
        return FDO_OK;
    }
    uint8_t code_length_lengths[19] = {0};
    for (size_t i = 0; i < d->hclen; i++) {
        code_length_lengths[CLCL_ORDER[i]] = (uint8_t)peek_bits(&d->bits, 3);
        consume_bits(&d->bits, 3);
        if (i == 17) {
            fill_buffer(&d->bits, remaining_input);
        }
    }
    uint16_t codes[19] = {0};
    uint16_t sec_dummy[1];
    size_t sec_len = 0;
    if (!fdo_build_table(code_length_lengths, 19, NULL, 0, codes, d->cl_table, 128, sec_dummy, 0,
                         &sec_len, 0, 0)) {
        return FDO_BAD_CODE_LENGTH_HUFFMAN_TREE;
    }
    d->state = ST_CODE_LENGTHS;
    d->num_lengths_read = 0;
    return FDO_OK;
}

/* read_code_lengths: src/decompress.rs:479-555 */
static int read_code_lengths(fdo_decompressor *d, Slice *remaining_input) {
    size_t total_lengths = d->hlit + d->hdist;
    while (d->num_lengths_read < total_lengths) {
        fill_buffer(&d->bits, remaining_input);
        if (d->bits.nbits < 7) {
            return FDO_OK;
        }
        uint64_t code = peek_bits(&d->bits, 7);
        uint32_t entry = d->cl_table[code];
        uint8_t length = (uint8_t)(entry & 0x7);
        uint8_t symbol = (uint8_t)(entry >> 16);

        if (symbol <= 15) {
            d->code_lengths[d->num_lengths_read] = symbol;
            d->num_lengths_read += 1;
            consume_bits(&d->bits, length);
        } else {
            size_t base_repeat;
            uint8_t extra_bits;
            if (symbol == 16) {
                base_repeat = 3;
                extra_bits = 2;
            } else if (symbol == 17) {
                base_repeat = 3;
                extra_bits = 3;
            } else {
                base_repeat = 11;
                extra_bits = 7;
            }
            if (d->bits.nbits < (uint8_t)(length + extra_bits)) {
                return FDO_OK;
            }
            uint8_t value = 0;
            if (symbol == 16) {
                if (d->num_lengths_read == 0) {
                    return FDO_INVALID_CODE_LENGTH_REPEAT;
                }
                value = d->code_lengths[d->num_lengths_read - 1];
            }
            size_t repeat =
                (size_t)(peek_bits(&d->bits, (uint8_t)(length + extra_bits)) >> length) + base_repeat;
            if (d->num_lengths_read + repeat > total_lengths) {
                return FDO_INVALID_CODE_LENGTH_REPEAT;
            }
            for (size_t i = 0; i < repeat; i++) {
                d->code_lengths[d->num_lengths_read + i] = value;
            }
            d->num_lengths_read += repeat;
            consume_bits(&d->bits, (uint8_t)(length + extra_bits));
        }
    }

    memmove(d->code_lengths + 288, d->code_lengths + d->hlit, total_lengths - d->hlit);
    for (size_t i = d->hlit; i < 288; i++) {
        d->code_lengths[i] = 0;
    }
    for (size_t i = 288 + d->hdist; i < 320; i++) {
        d->code_lengths[i] = 0;
    }
    int st = build_tables(d, d->hlit, d->code_lengths);
    if (st != FDO_OK) {
        return st;
    }
    d->state = ST_COMPRESSED_DATA;
    return FDO_OK;
}

/* Decompressor::read: src/decompress.rs:179-337 */
int fdo_decompressor_read(fdo_decompressor *d, const uint8_t *input, size_t input_len,
                          uint8_t *output, size_t output_len, size_t output_position,
                          size_t *consumed, size_t *produced) {
    *consumed = 0;
    *produced = 0;
    if (d->state == ST_DONE) {
        return FDO_OK;
    }
    if (output_position > output_len) {
        abort(); /* assert!(output_position <= output.len()) */
    }
    Slice remaining_input = {input, input_len};
    size_t output_index = output_position;

    /* :194-219 */
    if (d->q_kind != Q_NONE) {
        int kind = d->q_kind;
        d->q_kind = Q_NONE;
        size_t length = d->q_length;
        size_t room = output_len - output_index;
        size_t n = length < room ? length : room;
        if (kind == Q_RLE) {
            memset(output + output_index, d->q_data, n);
        } else {
            for (size_t i = 0; i < n; i++) {
                output[output_index + i] = output[output_index + i - d->q_dist];
            }
        }
        output_index += n;
        if (length - n != 0) {
            d->q_kind = kind;
            d->q_length = length - n;
            /* NB: returns before the trailing checksum.write (:331-333), exactly like
             * the reference's early `return Ok((0, n))`. */
            *consumed = 0;
            *produced = n;
            return FDO_OK;
        }
    }

    /* :221-329 */
    int have_last = 0;
    State last_state = ST_DONE;
    while (!have_last || last_state != d->state) {
        have_last = 1;
        last_state = d->state;
        int st = FDO_OK;
        switch (d->state) {
        case ST_ZLIB_HEADER: {
            fill_buffer(&d->bits, &remaining_input);
            if (d->bits.nbits < 16) {
                goto loop_done;
            }
            uint64_t input0 = peek_bits(&d->bits, 8);
            uint64_t input1 = (peek_bits(&d->bits, 16) >> 8) & 0xff;
            if ((input0 & 0x0f) != 0x08 || (input0 & 0xf0) > 0x70 || (input1 & 0x20) != 0 ||
                ((input0 << 8) | input1) % 31 != 0) {
                return FDO_BAD_ZLIB_HEADER;
            }
            consume_bits(&d->bits, 16);
            d->state = ST_BLOCK_HEADER;
            break;
        }
        case ST_BLOCK_HEADER:
            st = read_block_header(d, &remaining_input);
            if (st != FDO_OK) return st;
            break;
        case ST_CODE_LENGTH_CODES:
            st = read_code_length_codes(d, &remaining_input);
            if (st != FDO_OK) return st;
            break;
        case ST_CODE_LENGTHS:
            st = read_code_lengths(d, &remaining_input);
            if (st != FDO_OK) return st;
            break;
        case ST_COMPRESSED_DATA: {
            int block_status = MORE_DATA_PRESENT;
            st = read_compressed(d, &remaining_input, output, output_len, &output_index,
                                 &block_status);
            if (st != FDO_OK) return st;
            if (block_status == REACHED_END_OF_BLOCK) {
                d->state = d->last_block ? ST_CHECKSUM : ST_BLOCK_HEADER;
            }
            break;
        }
        case ST_UNCOMPRESSED_DATA: {
            while (d->bits.nbits > 0 && d->uncompressed_bytes_left > 0 &&
                   output_index < output_len) {
                output[output_index] = (uint8_t)peek_bits(&d->bits, 8);
                consume_bits(&d->bits, 8);
                output_index += 1;
                d->uncompressed_bytes_left -= 1;
            }
            if (d->bits.nbits == 0) {
                d->bits.buffer = 0;
            }
            size_t copy_bytes = d->uncompressed_bytes_left;
            if (remaining_input.len < copy_bytes) copy_bytes = remaining_input.len;
            if (output_len - output_index < copy_bytes) copy_bytes = output_len - output_index;
            memcpy(output + output_index, remaining_input.ptr, copy_bytes);
            remaining_input.ptr += copy_bytes;
            remaining_input.len -= copy_bytes;
            output_index += copy_bytes;
            d->uncompressed_bytes_left = (uint16_t)(d->uncompressed_bytes_left - copy_bytes);
            if (d->uncompressed_bytes_left == 0) {
                d->state = d->last_block ? ST_CHECKSUM : ST_BLOCK_HEADER;
            }
            break;
        }
        case ST_CHECKSUM: {
            fill_buffer(&d->bits, &remaining_input);
            uint8_t align_bits = d->bits.nbits % 8;
            if (d->bits.nbits >= 32 + align_bits) {
                d->checksum = fdo_adler32_update(d->checksum, output + output_position,
                                                 output_index - output_position);
                if (align_bits != 0) {
                    consume_bits(&d->bits, align_bits);
                }
                uint32_t stored = (uint32_t)peek_bits(&d->bits, 32);
                stored = ((stored & 0xff) << 24) | ((stored & 0xff00) << 8) |
                         ((stored >> 8) & 0xff00) | (stored >> 24); /* swap_bytes */
                if (!d->ignore_adler32 && stored != d->checksum) {
                    return FDO_WRONG_CHECKSUM;
                }
                d->state = ST_DONE;
                consume_bits(&d->bits, 32);
                goto loop_done;
            }
            break;
        }
        case ST_DONE:
            abort(); /* unreachable!() */
        }
    }
loop_done:
    /* :331-333 */
    if (!d->ignore_adler32 && d->state != ST_DONE) {
        d->checksum = fdo_adler32_update(d->checksum, output + output_position,
                                         output_index - output_position);
    }
    *consumed = input_len - remaining_input.len;
    *produced = output_index - output_position;
    return FDO_OK;
}

/* decompress_to_vec_bounded: src/decompress.rs:1111-1144 */
int fdo_decompress_bounded(const uint8_t *input, size_t input_len, uint8_t *out, size_t maxlen,
                           size_t *out_len, int ignore_adler32, uint32_t *adler) {
    fdo_decompressor *d = fdo_decompressor_new();
    if (ignore_adler32) {
        fdo_decompressor_ignore_adler32(d);
    }
    /* `output` is the Vec: the caller's buffer is its backing store (capacity maxlen),
     * `vec_len` its current len(); newly exposed bytes are zeroed like Vec::resize. */
    size_t vec_len = maxlen < 1024 ? maxlen : 1024;
    memset(out, 0, vec_len);
    size_t input_index = 0, output_index = 0;
    int status = FDO_OK;
    for (;;) {
        size_t consumed, produced;
        status = fdo_decompressor_read(d, input + input_index, input_len - input_index, out,
                                       vec_len, output_index, &consumed, &produced);
        if (status != FDO_OK) {
            break;
        }
        input_index += consumed;
        output_index += produced;
        if (fdo_decompressor_is_done(d)) {
            break;
        } else if (output_index == maxlen) {
            status = FDO_OUTPUT_TOO_LARGE;
            break;
        } else if (output_index == vec_len) {
            size_t new_len = output_index + 32 * 1024;
            if (new_len > maxlen) new_len = maxlen;
            memset(out + vec_len, 0, new_len - vec_len);
            vec_len = new_len;
            continue;
        } else if (input_index == input_len) {
            status = FDO_INSUFFICIENT_INPUT;
            break;
        } else {
            abort(); /* unreachable!("Read() call violated post-condition") */
        }
    }
    *out_len = output_index;
    if (adler) {
        *adler = fdo_adler32(out, output_index);
    }
    fdo_decompressor_free(d);
    return status;
}

/* decompress_by_chunks: src/decompress/tests/test_utils.rs:47-87 */
int fdo_decompress_by_chunks(const uint8_t *input, size_t input_len, long chunk, uint8_t *out,
                             size_t out_cap, size_t *out_len) {
    fdo_decompressor *d = fdo_decompressor_new();
    fdo_decompressor_ignore_adler32(d);
    size_t in_pos = 0, out_pos = 0;
    int iteration_counter = 0;
    int first = 1;
    int result = FDO_OK;
    while (!fdo_decompressor_is_done(d)) {
        iteration_counter += 1;
        if (iteration_counter > 5000) {
            result = -2;
            break;
        }
        size_t chunk_size;
        if (chunk <= 0) {
            chunk_size = first ? input_len : 0; /* vec![input.len()] then unwrap_or(0) */
        } else {
            chunk_size = (size_t)chunk; /* iter::repeat(chunk) */
        }
        first = 0;
        size_t start = in_pos;
        size_t end = start + chunk_size < input_len ? start + chunk_size : input_len;
        size_t consumed, produced;
        int st = fdo_decompressor_read(d, input + start, end - start, out, out_cap, out_pos,
                                       &consumed, &produced);
        if (st != FDO_OK) {
            result = st;
            break;
        }
        in_pos += consumed;
        out_pos += produced;
        if (out_pos == out_cap && consumed == 0 && !fdo_decompressor_is_done(d)) {
            result = -1;
            break;
        }
    }
    *out_len = out_pos;
    fdo_decompressor_free(d);
    return result;
}

/* ------------------------------------------------------------------------- */
/* compress/ultrafast.rs                                                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint32_t checksum;
    uint64_t buffer;
    uint8_t nbits;
    uint8_t *out;
    size_t pos, cap;
} UltraFast; /* src/compress/ultrafast.rs:9-14 */

static void uf_write_all(UltraFast *c, const uint8_t *p, size_t n) {
    if (c->pos + n > c->cap) {
        abort();
    }
    memcpy(c->out + c->pos, p, n);
    c->pos += n;
}

/* write_bits: src/compress/ultrafast.rs:16-29 */
static void uf_write_bits(UltraFast *c, uint64_t bits, uint8_t nbits) {
    c->buffer |= bits << c->nbits;
    c->nbits = (uint8_t)(c->nbits + nbits);
    if (c->nbits >= 64) {
        uint8_t le[8];
        memcpy(le, &c->buffer, 8);
        uf_write_all(c, le, 8);
        c->nbits = (uint8_t)(c->nbits - 64);
        unsigned sh = (unsigned)(nbits - c->nbits);
        c->buffer = sh < 64 ? bits >> sh : 0; /* checked_shr(..).unwrap_or(0) */
    }
}

/* flush: src/compress/ultrafast.rs:31-43 */
static void uf_flush(UltraFast *c) {
    if (c->nbits % 8 != 0) {
        uf_write_bits(c, 0, (uint8_t)(8 - c->nbits % 8));
    }
    if (c->nbits > 0) {
        uint8_t le[8];
        memcpy(le, &c->buffer, 8);
        uf_write_all(c, le, c->nbits / 8);
        c->buffer = 0;
        c->nbits = 0;
    }
}

/* write_run: src/compress/ultrafast.rs:45-67 */
static void uf_write_run(UltraFast *c, uint32_t run) {
    uf_write_bits(c, HUFFMAN_CODES[0], HUFFMAN_LENGTHS[0]);
    run -= 1;
    while (run >= 258) {
        uf_write_bits(c, HUFFMAN_CODES[285], (uint8_t)(HUFFMAN_LENGTHS[285] + 1));
        run -= 258;
    }
    if (run > 4) {
        unsigned sym = LENGTH_TO_SYMBOL[run - 3];
        uf_write_bits(c, HUFFMAN_CODES[sym], HUFFMAN_LENGTHS[sym]);
        uint8_t len_extra = LENGTH_TO_LEN_EXTRA[run - 3];
        uint64_t extra = (run - 3) & ((1u << len_extra) - 1); /* BITMASKS[len_extra] */
        uf_write_bits(c, extra, (uint8_t)(len_extra + 1));
    } else {
        uf_write_bits(c, 0, (uint8_t)(run * HUFFMAN_LENGTHS[0]));
    }
}

static unsigned tz64(uint64_t v) { return (unsigned)__builtin_ctzll(v); }
static unsigned lz64(uint64_t v) { return (unsigned)__builtin_clzll(v); }

size_t fdo_ultrafast_bound(size_t len) {
    /* 53 header bytes + ceil((5 + 12*len + 12) / 8) + 4 */
    return 53 + (5 + 12 * len + 12 + 7) / 8 + 4;
}

/* new + write_headers + write_data + finish: src/compress/ultrafast.rs:70-181 */
size_t fdo_compress_ultra_fast(const uint8_t *data, size_t len, uint8_t *out, size_t out_cap) {
    ensure_tables();
    UltraFast c = {1, 0, 0, out, 0, out_cap};
    /* write_headers :81-91 */
    uf_write_all(&c, ULTRAFAST_HEADER, 53);
    uf_write_bits(&c, ULTRAFAST_HEADER[53], 5);

    /* write_data :94-167 */
    c.checksum = fdo_adler32_update(c.checksum, data, len);
    uint32_t run = 0;
    size_t nchunks = len / 8;
    for (size_t ci = 0; ci < nchunks; ci++) {
        const uint8_t *chunk = data + ci * 8;
        uint64_t ichunk;
        memcpy(&ichunk, chunk, 8);
        if (ichunk == 0) {
            run += 8;
            continue;
        } else if (run > 0) {
            uint32_t run_extra = tz64(ichunk) / 8;
            uf_write_run(&c, run + run_extra);
            run = 0;
            if (run_extra > 0) {
                run = lz64(ichunk) / 8;
                for (size_t k = run_extra; k < 8 - run; k++) {
                    uint8_t b = chunk[k];
                    uf_write_bits(&c, HUFFMAN_CODES[b], HUFFMAN_LENGTHS[b]);
                }
                continue;
            }
        }
        uint32_t run_start = lz64(ichunk) / 8;
        if (run_start > 0) {
            for (size_t k = 0; k < 8 - run_start; k++) {
                uint8_t b = chunk[k];
                uf_write_bits(&c, HUFFMAN_CODES[b], HUFFMAN_LENGTHS[b]);
            }
            run = run_start;
            continue;
        }
        uint8_t n0 = HUFFMAN_LENGTHS[chunk[0]], n1 = HUFFMAN_LENGTHS[chunk[1]];
        uint8_t n2 = HUFFMAN_LENGTHS[chunk[2]], n3 = HUFFMAN_LENGTHS[chunk[3]];
        uint64_t bits = (uint64_t)HUFFMAN_CODES[chunk[0]] |
                        ((uint64_t)HUFFMAN_CODES[chunk[1]] << n0) |
                        ((uint64_t)HUFFMAN_CODES[chunk[2]] << (n0 + n1)) |
                        ((uint64_t)HUFFMAN_CODES[chunk[3]] << (n0 + n1 + n2));
        uf_write_bits(&c, bits, (uint8_t)(n0 + n1 + n2 + n3));
        uint8_t n4 = HUFFMAN_LENGTHS[chunk[4]], n5 = HUFFMAN_LENGTHS[chunk[5]];
        uint8_t n6 = HUFFMAN_LENGTHS[chunk[6]], n7 = HUFFMAN_LENGTHS[chunk[7]];
        uint64_t bits2 = (uint64_t)HUFFMAN_CODES[chunk[4]] |
                         ((uint64_t)HUFFMAN_CODES[chunk[5]] << n4) |
                         ((uint64_t)HUFFMAN_CODES[chunk[6]] << (n4 + n5)) |
                         ((uint64_t)HUFFMAN_CODES[chunk[7]] << (n4 + n5 + n6));
        uf_write_bits(&c, bits2, (uint8_t)(n4 + n5 + n6 + n7));
    }
    if (run > 0) {
        uf_write_run(&c, run);
    }
    for (size_t k = nchunks * 8; k < len; k++) {
        uint8_t b = data[k];
        uf_write_bits(&c, HUFFMAN_CODES[b], HUFFMAN_LENGTHS[b]);
    }

    /* finish :170-181 */
    uf_write_bits(&c, HUFFMAN_CODES[256], HUFFMAN_LENGTHS[256]);
    uf_flush(&c);
    uint8_t be[4] = {(uint8_t)(c.checksum >> 24), (uint8_t)(c.checksum >> 16),
                     (uint8_t)(c.checksum >> 8), (uint8_t)c.checksum};
    uf_write_all(&c, be, 4);
    return c.pos;
}

/* Compressor level 0 via compress_to_vec_with_level(input, 0):
 * src/compress/mod.rs:69-71 (header 78 01), :126-160 (write_data compresses directly
 * with Flush::None), :194-214 (finish with Flush::Finish), :234-268 (stored blocks;
 * empty remaining input at Finish is `write_bits(3, 10)`). */
size_t fdo_compress_stored(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap) {
    size_t pos = 0;
#define PUT(b)                                   \
    do {                                         \
        if (pos >= out_cap) abort();             \
        out[pos++] = (uint8_t)(b);               \
    } while (0)
    PUT(0x78);
    PUT(0x01);
    const size_t MAXB = 65535;
    size_t off = 0;
    /* write_data -> compress(Flush::None): full 65535-byte blocks only */
    while (len - off > MAXB) {
        PUT(0x00); /* write_bits(0,3) + flush => one zero byte */
        PUT(0xff);
        PUT(0xff);
        PUT(0x00);
        PUT(0x00);
        if (pos + MAXB > out_cap) abort();
        memcpy(out + pos, input + off, MAXB);
        pos += MAXB;
        off += MAXB;
    }
    if (len - off == MAXB) { /* :252 input.len() == STORED_BLOCK_MAX_SIZE with Flush::None */
        PUT(0x00);
        PUT(0xff);
        PUT(0xff);
        PUT(0x00);
        PUT(0x00);
        if (pos + MAXB > out_cap) abort();
        memcpy(out + pos, input + off, MAXB);
        pos += MAXB;
        off += MAXB;
    }
    /* finish -> compress(Flush::Finish) */
    if (len - off == 0) {
        PUT(0x03); /* write_bits(3, 10) then flush: 0x03 0x00 */
        PUT(0x00);
    } else {
        size_t n = len - off;
        PUT(0x01);
        PUT(n & 0xff);
        PUT((n >> 8) & 0xff);
        PUT((~n) & 0xff);
        PUT(((~n) >> 8) & 0xff);
        if (pos + n > out_cap) abort();
        memcpy(out + pos, input + off, n);
        pos += n;
    }
    uint32_t ck = fdo_adler32(input, len);
    PUT(ck >> 24);
    PUT(ck >> 16);
    PUT(ck >> 8);
    PUT(ck);
#undef PUT
    return pos;
}

/* ------------------------------------------------------------------------- */
/* Batch helpers (CPU baseline leg; one stream per task)                      */
/* ------------------------------------------------------------------------- */

typedef struct {
    const uint8_t *in;
    const uint64_t *in_off;
    uint8_t *out;
    const uint64_t *out_off;
    uint32_t *out_len, *status, *adler;
    uint64_t n;
    int ignore_adler32, encode;
    volatile uint64_t *next;
} BatchJob;

static void *batch_worker(void *arg) {
    BatchJob *j = (BatchJob *)arg;
    for (;;) {
        uint64_t i = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (i >= j->n) {
            break;
        }
        const uint8_t *src = j->in + j->in_off[i];
        size_t src_len = (size_t)(j->in_off[i + 1] - j->in_off[i]);
        uint8_t *dst = j->out + j->out_off[i];
        size_t cap = (size_t)(j->out_off[i + 1] - j->out_off[i]);
        if (j->encode) {
            j->out_len[i] = (uint32_t)fdo_compress_ultra_fast(src, src_len, dst, cap);
        } else {
            size_t n = 0;
            uint32_t ad = 0;
            int st = fdo_decompress_bounded(src, src_len, dst, cap, &n, j->ignore_adler32, &ad);
            j->out_len[i] = (uint32_t)n;
            j->status[i] = (uint32_t)st;
            if (j->adler) {
                j->adler[i] = ad;
            }
        }
    }
    return NULL;
}

static void run_batch(BatchJob *job, int nthreads) {
    volatile uint64_t next = 0;
    job->next = &next;
    if (nthreads <= 1) {
        batch_worker(job);
        return;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    int started = 0;
    if (th) {
        for (int t = 0; t < nthreads; t++) {
            if (pthread_create(&th[started], NULL, batch_worker, job) != 0) {
                break; /* fewer threads than asked for: the ones that exist share the work */
            }
            started++;
        }
    }
    if (started == 0) {
        batch_worker(job);
    }
    for (int t = 0; t < started; t++) {
        pthread_join(th[t], NULL);
    }
    free(th);
}

void fdo_inflate_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                       const uint64_t *out_off, uint32_t *out_len, uint32_t *status,
                       uint32_t *adler, uint64_t n, int ignore_adler32, int nthreads) {
    ensure_tables();
    BatchJob job = {in, in_off, out, out_off, out_len, status, adler, n, ignore_adler32, 0, NULL};
    run_batch(&job, nthreads);
}

void fdo_deflate_ultrafast_batch(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                                 const uint64_t *out_off, uint32_t *out_len, uint64_t n,
                                 int nthreads) {
    ensure_tables();
    BatchJob job = {in, in_off, out, out_off, out_len, NULL, NULL, n, 0, 1, NULL};
    run_batch(&job, nthreads);
}

/* ------------------------------------------------------------------------- */
/* Timed CPU baseline (bench.py cpu_baseline leg)                             */
/* ------------------------------------------------------------------------- */
/* Threads are created ONCE; thread t decodes streams t, t+T, t+2T, ... of the sample `passes`
 * times, so thread start-up is outside the measurement (first barrier) and every thread has
 * passes * n / T streams of work.  kind 0 = this oracle (the port of the reference algorithm),
 * kind 1 = system zlib `uncompress` (second comparator). */
#include <time.h>
#include <zlib.h>

typedef struct {
    const uint8_t *in;
    const uint64_t *in_off;
    uint8_t *out;
    const uint64_t *out_off;
    uint64_t n;
    int kind, passes, nthreads;
    pthread_barrier_t *bar;
    uint64_t *errors;
} TimedShared;

typedef struct {
    TimedShared *sh;
    int tid;
} TimedArg;

static void *timed_worker(void *arg) {
    TimedArg *ta = (TimedArg *)arg;
    TimedShared *j = ta->sh;
    uint64_t bad = 0;
    pthread_barrier_wait(j->bar); /* start line */
    for (int p = 0; p < j->passes; p++) {
        for (uint64_t i = (uint64_t)ta->tid; i < j->n; i += (uint64_t)j->nthreads) {
            const uint8_t *src = j->in + j->in_off[i];
            size_t src_len = (size_t)(j->in_off[i + 1] - j->in_off[i]);
            uint8_t *dst = j->out + j->out_off[i];
            size_t cap = (size_t)(j->out_off[i + 1] - j->out_off[i]);
            if (j->kind == 0) {
                size_t n = 0;
                uint32_t ad = 0;
                int st = fdo_decompress_bounded(src, src_len, dst, cap, &n, 0, &ad);
                bad += (st != FDO_OK);
            } else {
                uLongf dl = (uLongf)cap;
                bad += (uncompress(dst, &dl, src, (uLong)src_len) != Z_OK);
            }
        }
    }
    pthread_barrier_wait(j->bar); /* finish line */
    __atomic_fetch_add(j->errors, bad, __ATOMIC_RELAXED);
    return NULL;
}

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Returns the seconds between the start and the finish line, < 0 on failure (threads could not be
 * created / streams failed to decode). */
double fdo_timed_inflate(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                         const uint64_t *out_off, uint64_t n, int nthreads, int passes, int kind) {
    ensure_tables();
    if (nthreads < 1) {
        nthreads = 1;
    }
    pthread_barrier_t bar;
    uint64_t errors = 0;
    if (pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1) != 0) {
        return -1.0;
    }
    TimedShared sh = {in, in_off, out, out_off, n, kind, passes, nthreads, &bar, &errors};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    TimedArg *args = (TimedArg *)malloc(sizeof(TimedArg) * (size_t)nthreads);
    if (!th || !args) {
        free(th);
        free(args);
        pthread_barrier_destroy(&bar);
        return -1.0;
    }
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        args[t].sh = &sh;
        args[t].tid = t;
        if (pthread_create(&th[t], NULL, timed_worker, &args[t]) != 0) {
            break;
        }
        started++;
    }
    double dt = -1.0;
    if (started == nthreads) {
        pthread_barrier_wait(&bar);
        double t0 = now_seconds();
        pthread_barrier_wait(&bar);
        dt = now_seconds() - t0;
    } else {
        /* cannot release a barrier sized for more threads than exist: cancel the ones waiting */
        for (int t = 0; t < started; t++) {
            pthread_cancel(th[t]);
        }
    }
    for (int t = 0; t < started; t++) {
        pthread_join(th[t], NULL);
    }
    free(th);
    free(args);
    pthread_barrier_destroy(&bar);
    if (errors != 0) {
        return -2.0;
    }
    return dt;
}

/* ========================================================================= */
/* General encoder, levels 1 ("fast") and RLE                                 */
/*   Compressor            src/compress/mod.rs:47-217                         */
/*   GreedyParser          src/compress/parse/greedy.rs:10-92                 */
/*   RleParser             src/compress/parse/rle.rs:5-48                     */
/*   ParserInner           src/compress/parse/mod.rs:18-181                   */
/*   HashTableMatchFinder  src/compress/matchfinder/hashtable.rs:5-63         */
/*   match_length/rle_match src/compress/matchfinder/mod.rs:42-145            */
/*   write_block / build_huffman_tree  src/compress/bitstream.rs:41-325       */
/*   BitWriter             src/compress/bitwriter.rs:3-51                     */
/*                                                                            */
/* PARITY PIN of this section: the reference holds no golden compressed bytes  */
/* for these encoders (only round-trip tests, src/decompress.rs:1235-1259, and */
/* the empty-input bytes 78 01 03 00 00 00 00 01, src/compress/mod.rs:71,      */
/* 234-238), and it cannot be built here.  Pinned by restatement + the KAT +   */
/* zlib round trips (tests/test_oracle_golden.py).  Two library behaviours the */
/* reference inherits from Rust's std are restated as the std documents /      */
/* implements them: BinaryHeap (rebuild, pop = sift_down_to_bottom + sift_up,  */
/* PeekMut drop = sift_down) for tie-breaking in build_huffman_tree, and       */
/* `sort_unstable_by_key` in the length-limiting branch, which is an insertion */
/* sort (hence stable) for the <= 20-element code-length alphabet; for the     */
/* 286 / 30-element alphabets that branch needs a code deeper than 15 bits and */
/* its tie order is implementation-defined in the reference itself -- a stable */
/* order is used here.                                                        */
/* ========================================================================= */

typedef struct {
    uint64_t buffer;
    uint8_t nbits;
    uint8_t *out;
    size_t cap, pos;
    int overflow;
} GBitWriter;

static void gbw_raw(GBitWriter *w, const uint8_t *p, size_t n) {
    if (w->pos + n > w->cap) {
        w->overflow = 1;
        return;
    }
    memcpy(w->out + w->pos, p, n);
    w->pos += n;
}

/* bitwriter.rs:17-31 */
static void gbw_write_bits(GBitWriter *w, uint64_t bits, uint8_t nbits) {
    w->buffer |= bits << w->nbits;
    w->nbits = (uint8_t)(w->nbits + nbits);
    if (w->nbits >= 64) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) {
            b[i] = (uint8_t)(w->buffer >> (8 * i));
        }
        gbw_raw(w, b, 8);
        w->nbits = (uint8_t)(w->nbits - 64);
        unsigned sh = (unsigned)(nbits - w->nbits);
        w->buffer = sh >= 64 ? 0 : bits >> sh; /* checked_shr(..).unwrap_or(0) */
    }
}

/* bitwriter.rs:33-45 */
static void gbw_flush(GBitWriter *w) {
    if (w->nbits % 8 != 0) {
        gbw_write_bits(w, 0, (uint8_t)(8 - w->nbits % 8));
    }
    if (w->nbits > 0) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) {
            b[i] = (uint8_t)(w->buffer >> (8 * i));
        }
        gbw_raw(w, b, w->nbits / 8u);
        w->buffer = 0;
        w->nbits = 0;
    }
}

/* ---- build_huffman_tree (bitstream.rs:198-325) ---- */
typedef struct {
    uint32_t f;
    uint16_t idx;
} HItem;
/* Ord for Item: `other.0.cmp(&self.0)` (bitstream.rs:219-223): a <= b  <=>  a.f >= b.f */
static int hi_le(HItem a, HItem b) { return a.f >= b.f; }
static int hi_ge(HItem a, HItem b) { return a.f <= b.f; }
static int hi_lt(HItem a, HItem b) { return a.f > b.f; }

/* std::collections::BinaryHeap::sift_down_range */
static void heap_sift_down_range(HItem *d, size_t pos, size_t end) {
    HItem elem = d[pos];
    size_t hole = pos, child = 2 * hole + 1;
    size_t lim = end >= 2 ? end - 2 : 0;
    while (child <= lim) {
        if (hi_le(d[child], d[child + 1])) {
            child++;
        }
        if (hi_ge(elem, d[child])) {
            d[hole] = elem;
            return;
        }
        d[hole] = d[child];
        hole = child;
        child = 2 * hole + 1;
    }
    if (child == end - 1 && hi_lt(elem, d[child])) {
        d[hole] = d[child];
        hole = child;
    }
    d[hole] = elem;
}
/* std::collections::BinaryHeap::pop = swap with the last, sift_down_to_bottom(0), sift_up */
static HItem heap_pop(HItem *d, size_t *len) {
    HItem item = d[*len - 1];
    (*len)--;
    if (*len > 0) {
        HItem t = d[0];
        d[0] = item;
        item = t;
        size_t end = *len, hole = 0, child = 1;
        HItem elem = d[0];
        size_t lim = end >= 2 ? end - 2 : 0;
        while (child <= lim) {
            if (hi_le(d[child], d[child + 1])) {
                child++;
            }
            d[hole] = d[child];
            hole = child;
            child = 2 * hole + 1;
        }
        if (child == end - 1) {
            d[hole] = d[child];
            hole = child;
        }
        /* sift_up(0, hole) */
        while (hole > 0) {
            size_t parent = (hole - 1) / 2;
            if (hi_le(elem, d[parent])) {
                break;
            }
            d[hole] = d[parent];
            hole = parent;
        }
        d[hole] = elem;
    }
    return item;
}


/* How often build_huffman_tree had to shorten a tree (bitstream.rs:262-305): [0] limit 15, [1] limit 7.
 * Test instrumentation only (not thread-safe): lets a test show that its input takes that path. */
unsigned long fdo_length_limit_events[2];

static int g_build_huffman_tree(const uint32_t *freq, size_t n, uint8_t *lengths, uint16_t *codes,
                                uint8_t length_limit) {
    size_t used = 0, first = 0;
    for (size_t i = 0; i < n; i++) {
        if (freq[i] > 0) {
            if (used == 0) {
                first = i;
            }
            used++;
        }
    }
    memset(lengths, 0, n);
    memset(codes, 0, 2 * n);
    if (used <= 1) { /* :206-213 */
        if (used == 1) {
            lengths[first] = 1;
        }
        return 0;
    }
    HItem heap[286];
    uint16_t in_left[286], in_right[286];
    size_t hl = 0, ni = 0;
    for (size_t i = 0; i < n; i++) {
        if (freq[i] > 0) {
            heap[hl].f = freq[i];
            heap[hl].idx = (uint16_t)i;
            hl++;
        }
    }
    for (size_t k = hl / 2; k > 0;) { /* BinaryHeap::from(vec): rebuild */
        k--;
        heap_sift_down_range(heap, k, hl);
    }
    while (hl > 1) { /* :236-244 */
        HItem a = heap_pop(heap, &hl);
        in_left[ni] = a.idx;
        in_right[ni] = heap[0].idx;
        ni++;
        heap[0].f = a.f + heap[0].f;
        heap[0].idx = (uint16_t)(ni + n - 1);
        heap_sift_down_range(heap, 0, hl); /* PeekMut::drop */
    }
    /* :247-259 walk the tree */
    struct {
        uint16_t node;
        int depth;
    } stack[600];
    size_t sp = 0;
    stack[sp].node = heap[0].idx;
    stack[sp].depth = 0;
    sp++;
    while (sp > 0) {
        sp--;
        uint16_t node = stack[sp].node;
        int depth = stack[sp].depth;
        if (node < n) {
            lengths[node] = (uint8_t)depth;
        } else {
            stack[sp].node = in_left[node - n];
            stack[sp].depth = depth + 1;
            sp++;
            stack[sp].node = in_right[node - n];
            stack[sp].depth = depth + 1;
            sp++;
        }
    }
    /* :262-305 limit the lengths */
    uint8_t max_length = 0;
    for (size_t i = 0; i < n; i++) {
        if (lengths[i] > max_length) {
            max_length = lengths[i];
        }
    }
    if (max_length > length_limit) {
        fdo_length_limit_events[length_limit == 7 ? 1 : 0]++;
        uint32_t counts[16] = {0};
        for (size_t i = 0; i < n; i++) {
            counts[lengths[i] < length_limit ? lengths[i] : length_limit]++;
        }
        uint32_t total = 0;
        for (unsigned i = 1; i <= length_limit; i++) {
            total += counts[i] << (length_limit - i);
        }
        while (total > (1u << length_limit)) {
            unsigned i = length_limit - 1u;
            while (counts[i] == 0) {
                i--;
            }
            counts[i]--;
            counts[length_limit]--;
            counts[i + 1] += 2;
            total--;
        }
        /* sort_unstable_by_key(frequency): stable order restated (see the section header) */
        uint16_t order[286];
        for (size_t i = 0; i < n; i++) {
            order[i] = (uint16_t)i;
        }
        for (size_t i = 1; i < n; i++) {
            uint16_t v = order[i];
            size_t j = i;
            while (j > 0 && freq[order[j - 1]] > freq[v]) {
                order[j] = order[j - 1];
                j--;
            }
            order[j] = v;
        }
        uint8_t len = length_limit;
        for (size_t k = 0; k < n; k++) {
            size_t i = order[k];
            if (freq[i] > 0) {
                while (counts[len] == 0) {
                    len--;
                }
                lengths[i] = len;
                counts[len]--;
            }
        }
    }
    /* :308-320 canonical codes, bit-reversed */
    uint32_t code = 0;
    for (unsigned len = 1; len <= length_limit; len++) {
        for (size_t i = 0; i < n; i++) {
            if (lengths[i] == len) {
                uint16_t c = (uint16_t)code, r = 0;
                for (int b = 0; b < 16; b++) {
                    r = (uint16_t)((r << 1) | ((c >> b) & 1));
                }
                codes[i] = (uint16_t)(r >> (16 - len));
                code++;
            }
        }
        code <<= 1;
    }
    return 1;
}

/* Symbol (bitstream.rs:29-39): a literal run [start, end) or a back-reference */
typedef struct {
    uint32_t a; /* LiteralRun: start          | Backref: length | 0x80000000 */
    uint32_t b; /* LiteralRun: end            | Backref: distance | dist_sym << 16 */
} GSymbol;
#define GSYM_IS_BACKREF(s) (((s).a & 0x80000000u) != 0)

/* distance_to_dist_sym (bitstream.rs:16-27) */
static uint8_t g_distance_to_dist_sym(uint16_t distance) {
    static const uint8_t LOOKUP[16] = {0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7};
    if (distance <= 16) {
        return LOOKUP[distance - 1];
    }
    uint8_t dist_sym = 29;
    while (dist_sym > 0 && distance < DIST_SYM_TO_DIST_BASE[dist_sym]) {
        dist_sym--;
    }
    return dist_sym;
}

static const uint32_t G_BITMASKS[17] = {0x0000, 0x0001, 0x0003, 0x0007, 0x000F, 0x001F, 0x003F, 0x007F, 0x00FF,
                                        0x01FF, 0x03FF, 0x07FF, 0x0FFF, 0x1FFF, 0x3FFF, 0x7FFF, 0xFFFF};

/* write_block (bitstream.rs:41-195) */
static void g_write_block(GBitWriter *w, const uint8_t *data, uint32_t base_index, const GSymbol *symbols,
                          size_t nsym, int eof) {
    uint32_t frequencies[286] = {0}, dist_frequencies[30] = {0};
    frequencies[256] = 1;
    for (size_t k = 0; k < nsym; k++) { /* the f2/f3/f4 split of :48-78 only reorders the additions */
        if (GSYM_IS_BACKREF(symbols[k])) {
            unsigned length = symbols[k].a & 0xFFFF, dist_sym = symbols[k].b >> 16;
            frequencies[LENGTH_TO_SYMBOL[length - 3]]++;
            dist_frequencies[dist_sym]++;
        } else {
            for (uint32_t p = symbols[k].a - base_index; p < symbols[k].b - base_index; p++) {
                frequencies[data[p]]++;
            }
        }
    }
    uint8_t lengths[286], dist_lengths[30], cl_lengths[19];
    uint16_t codes[286], dist_codes[30], cl_codes[19];
    g_build_huffman_tree(frequencies, 286, lengths, codes, 15);
    g_build_huffman_tree(dist_frequencies, 30, dist_lengths, dist_codes, 15);
    size_t num_litlen_codes = 286, num_dist_codes = 30;
    while (num_litlen_codes > 257 && lengths[num_litlen_codes - 1] == 0) {
        num_litlen_codes--;
    }
    while (num_dist_codes > 1 && dist_lengths[num_dist_codes - 1] == 0) {
        num_dist_codes--;
    }
    uint32_t cl_freq[19] = {0};
    for (size_t i = 0; i < num_litlen_codes; i++) {
        cl_freq[lengths[i]]++;
    }
    for (size_t i = 0; i < num_dist_codes; i++) {
        cl_freq[dist_lengths[i]]++;
    }
    g_build_huffman_tree(cl_freq, 19, cl_lengths, cl_codes, 7);

    gbw_write_bits(w, eof ? 5 : 4, 3); /* 0b101 / 0b100: BFINAL + dynamic */
    gbw_write_bits(w, num_litlen_codes - 257, 5);
    gbw_write_bits(w, num_dist_codes - 1, 5);
    gbw_write_bits(w, 15, 4);
    for (int j = 0; j < 19; j++) {
        gbw_write_bits(w, cl_lengths[CLCL_ORDER[j]], 3);
    }
    for (size_t i = 0; i < num_litlen_codes; i++) {
        gbw_write_bits(w, cl_codes[lengths[i]], cl_lengths[lengths[i]]);
    }
    for (size_t i = 0; i < num_dist_codes; i++) {
        gbw_write_bits(w, cl_codes[dist_lengths[i]], cl_lengths[dist_lengths[i]]);
    }
    for (size_t k = 0; k < nsym; k++) {
        if (GSYM_IS_BACKREF(symbols[k])) {
            unsigned length = symbols[k].a & 0xFFFF, distance = symbols[k].b & 0xFFFF, dist_sym = symbols[k].b >> 16;
            unsigned sym = LENGTH_TO_SYMBOL[length - 3];
            gbw_write_bits(w, codes[sym], lengths[sym]);
            uint8_t len_extra = LENGTH_TO_LEN_EXTRA[length - 3];
            gbw_write_bits(w, (length - 3) & G_BITMASKS[len_extra], len_extra);
            gbw_write_bits(w, dist_codes[dist_sym], dist_lengths[dist_sym]);
            gbw_write_bits(w, distance - DIST_SYM_TO_DIST_BASE[dist_sym], DIST_SYM_TO_DIST_EXTRA[dist_sym]);
        } else {
            /* groups of four literals in one write_bits (:134-160): the byte stream does not depend
             * on how the bits are grouped, but the 64-bit accumulator hand-over does not either */
            uint32_t p = symbols[k].a - base_index, end = symbols[k].b - base_index;
            for (; p + 4 <= end; p += 4) {
                uint8_t l0 = lengths[data[p]], l1 = lengths[data[p + 1]], l2 = lengths[data[p + 2]],
                        l3 = lengths[data[p + 3]];
                uint64_t v = (uint64_t)codes[data[p]] | ((uint64_t)codes[data[p + 1]] << l0) |
                             ((uint64_t)codes[data[p + 2]] << (l0 + l1)) | ((uint64_t)codes[data[p + 3]] << (l0 + l1 + l2));
                gbw_write_bits(w, v, (uint8_t)(l0 + l1 + l2 + l3));
            }
            for (; p < end; p++) {
                gbw_write_bits(w, codes[data[p]], lengths[data[p]]);
            }
        }
    }
    gbw_write_bits(w, codes[256], lengths[256]);
}

/* ---- match finders ---- */
typedef struct {
    uint16_t length, distance;
    size_t start;
} GMatch;
static GMatch gm_empty(void) {
    GMatch m = {0, 0, 0};
    return m;
}
static size_t gm_end(GMatch m) { return m.start + m.length; }

static uint64_t g_load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8); /* from_le_bytes / from_ne_bytes: little-endian hosts only (as the oracle's fill_buffer) */
    return v;
}
/* compute_hash (matchfinder/mod.rs:42-44) */
static uint32_t g_compute_hash(uint64_t v) { return (uint32_t)((11400714785074694791ull * v) >> 40); }

/* match_length::<true> (matchfinder/mod.rs:51-111) */
static void g_match_length8(uint64_t value, const uint8_t *data, size_t len, size_t anchor, size_t ip, size_t prev_index,
                            uint16_t *out_len, size_t *out_start) {
    uint64_t prev = g_load64(data + prev_index);
    if (value != prev) {
        *out_len = 0;
        *out_start = ip;
        return;
    }
    size_t length = 8;
    while (length < 258 && ip > anchor && prev_index > 0 && data[ip - 1] == data[prev_index - 1]) {
        length++;
        ip--;
        prev_index--;
    }
    size_t slice = len - ip - length;
    if (slice > 258 - length) {
        slice = 258 - length;
    }
    const uint8_t *a = data + ip + length, *b = data + prev_index + length;
    size_t k = 0;
    int done = 0;
    for (; k + 8 <= slice; k += 8) {
        uint64_t x = g_load64(a + k), y = g_load64(b + k);
        if (x == y) {
            length += 8;
        } else {
            length += (size_t)__builtin_ctzll(x ^ y) / 8;
            done = 1;
            break;
        }
    }
    if (!done) {
        for (; k < slice; k++) {
            if (a[k] != b[k]) {
                break;
            }
            length++;
        }
    }
    *out_len = (uint16_t)length;
    *out_start = ip;
}

/* rle_match (matchfinder/mod.rs:113-145) */
static GMatch g_rle_match(const uint8_t *data, size_t len, size_t last_match, size_t ip) {
    uint8_t value = data[ip];
    GMatch m = {4, 1, ip + 1};
    size_t min_start = 1;
    if (last_match > min_start) {
        min_start = last_match;
    }
    size_t e = gm_end(m);
    if (e > 258 && e - 258 > min_start) {
        min_start = e - 258;
    }
    while (m.start > min_start && data[m.start - 2] == value) {
        m.start--;
        m.length++;
    }
    const uint8_t *p = data + gm_end(m);
    size_t n = len - gm_end(m);
    if (n > (size_t)(258 - m.length)) {
        n = (size_t)(258 - m.length);
    }
    uint64_t v8 = 0x0101010101010101ull * value;
    size_t k = 0;
    for (; k + 8 <= n; k += 8) {
        uint64_t c = g_load64(p + k);
        if (c != v8) {
            m.length = (uint16_t)(m.length + __builtin_ctzll(c ^ v8) / 8);
            return m;
        }
        m.length += 8;
    }
    for (; k < n; k++) {
        if (p[k] != value) {
            break;
        }
        m.length++;
    }
    return m;
}

#define G_MAX_SYMBOLS (16384 + 8)
typedef struct {
    int use_hash;         /* 1: HashTableMatchFinder (level 1), 0: NullMatchFinder (RLE) */
    uint32_t *hash_table; /* CACHE_SIZE = 1 << 16 entries */
    uint8_t skip_ahead_shift;
    GSymbol *symbols;
    size_t nsym;
    size_t ip, last_match, last_block_end;
    uint32_t last_index;
    GMatch m; /* GreedyParser::m */
} GParser;

/* HashTableMatchFinder::get_and_insert (hashtable.rs:16-50) / NullMatchFinder */
static GMatch g_get_and_insert(GParser *ps, const uint8_t *data, size_t len, uint32_t base_index, size_t anchor,
                               size_t ip, uint64_t value) {
    if (!ps->use_hash) {
        return gm_empty();
    }
    uint32_t sub = (uint32_t)ip > 32768 ? (uint32_t)ip - 32768 : 0;
    uint32_t min_offset = base_index + sub;
    if (min_offset < 1) {
        min_offset = 1;
    }
    uint32_t hash_index = g_compute_hash(value) % 65536u;
    uint32_t offset = ps->hash_table[hash_index];
    ps->hash_table[hash_index] = (uint32_t)ip + base_index;
    if (offset >= min_offset) {
        uint16_t length;
        size_t start;
        g_match_length8(value, data, len, anchor, ip, (size_t)(offset - base_index), &length, &start);
        if (length >= 8) {
            GMatch m = {length, (uint16_t)(ip - (size_t)(offset - base_index)), start};
            return m;
        }
    }
    return gm_empty();
}

/* ParserInner::get_match (parse/mod.rs:58-85) */
static GMatch g_get_match(GParser *ps, const uint8_t *data, size_t len, uint32_t base_index, int fizzle) {
    uint64_t current = g_load64(data + ps->ip);
    if ((uint32_t)current == (uint32_t)(current >> 8)) {
        GMatch m = g_rle_match(data, len, ps->last_match, ps->ip);
        ps->ip = gm_end(m) - 3;
        return m;
    }
    size_t anchor = fizzle ? ps->ip : ps->last_match;
    GMatch m = g_get_and_insert(ps, data, len, base_index, anchor, ps->ip, current);
    if (fizzle) {
        while (m.length < 258 && m.start > ps->last_match && m.start > (size_t)m.distance + 1 &&
               data[m.start - 1] == data[m.start - m.distance - 1]) {
            m.length++;
            m.start--;
        }
    }
    ps->ip++;
    return m;
}

/* ParserInner::advance_to_match (parse/mod.rs:88-102) */
static GMatch g_advance_to_match(GParser *ps, const uint8_t *data, size_t len, uint32_t base_index, size_t max_ip) {
    while (ps->ip < max_ip) {
        GMatch m = g_get_match(ps, data, len, base_index, 0);
        if (m.length != 0) {
            return m;
        }
        ps->ip += (ps->ip - ps->last_match) >> ps->skip_ahead_shift;
    }
    return gm_empty();
}

/* ParserInner::advance (parse/mod.rs:105-114) */
static void g_advance(GParser *ps, const uint8_t *data, size_t len, uint32_t base_index, size_t end) {
    size_t stop = end < len - 8 ? end : len - 8;
    if (ps->use_hash) {
        for (size_t j = ps->ip; j < stop; j++) {
            ps->hash_table[g_compute_hash(g_load64(data + j)) % 65536u] = base_index + (uint32_t)j;
        }
    }
    if (end > ps->ip) {
        ps->ip = end;
    }
}

/* ParserInner::insert_match (parse/mod.rs:117-131) */
static void g_insert_match(GParser *ps, uint32_t base_index, GMatch m) {
    if (m.start > ps->last_match) {
        ps->symbols[ps->nsym].a = base_index + (uint32_t)ps->last_match;
        ps->symbols[ps->nsym].b = base_index + (uint32_t)m.start;
        ps->nsym++;
    }
    ps->symbols[ps->nsym].a = 0x80000000u | m.length;
    ps->symbols[ps->nsym].b = m.distance | ((uint32_t)g_distance_to_dist_sym(m.distance) << 16);
    ps->nsym++;
    ps->last_match = gm_end(m);
}

/* ParserInner::write_block_if_ready (parse/mod.rs:134-150) */
static void g_write_block_if_ready(GParser *ps, GBitWriter *w, const uint8_t *data, size_t len, uint32_t base_index,
                                   int finish) {
    if (ps->nsym >= 16384) {
        int last_block = finish && ps->last_match == len;
        g_write_block(w, data, base_index, ps->symbols, ps->nsym, last_block);
        ps->nsym = 0;
        ps->last_block_end = ps->last_match;
    }
}

/* start_compress (parse/mod.rs:46-55) + end_compress (:152-180).  flush: 0 None, 2 Finish. */
static size_t g_start_compress(GParser *ps, uint32_t base_index, size_t start) {
    uint32_t delta = base_index - ps->last_index;
    ps->ip -= delta;
    ps->last_match -= delta;
    ps->last_block_end = start;
    ps->last_index = base_index;
    return delta;
}
static size_t g_end_compress(GParser *ps, GBitWriter *w, const uint8_t *data, size_t len, uint32_t base_index,
                             size_t start, int finish) {
    if (finish && (ps->nsym != 0 || ps->last_match < len)) {
        if (ps->ip > len) {
            ps->ip = len;
        }
        if (ps->last_match < len) {
            ps->symbols[ps->nsym].a = base_index + (uint32_t)ps->last_match;
            ps->symbols[ps->nsym].b = base_index + (uint32_t)len;
            ps->nsym++;
            ps->ip = len;
            ps->last_match = len;
        }
        g_write_block(w, data, base_index, ps->symbols, ps->nsym, 1);
        ps->nsym = 0;
        ps->last_block_end = ps->ip;
    }
    return ps->last_block_end - start;
}

/* GreedyParser::compress (parse/greedy.rs:27-91) */
static size_t g_greedy_compress(GParser *ps, GBitWriter *w, const uint8_t *data, size_t len, uint32_t base_index,
                                size_t start, int finish) {
    size_t delta = g_start_compress(ps, base_index, start);
    if (ps->m.length != 0) {
        ps->m.start -= delta;
    }
    size_t lookahead = finish ? 7 : 258 + 8;
    size_t max_ip = len > lookahead ? len - lookahead : 0;
    for (;;) {
        if (ps->m.length == 0) {
            ps->m = g_advance_to_match(ps, data, len, base_index, max_ip);
            if (ps->m.length == 0) {
                break;
            }
        }
        g_advance(ps, data, len, base_index, gm_end(ps->m));
        GMatch m2 = gm_empty();
        if (ps->ip < max_ip) {
            m2 = g_get_match(ps, data, len, base_index, 1);
        } else if (!finish) {
            break;
        }
        if (m2.length == 0 || m2.start > ps->m.start + 1) {
            g_insert_match(ps, base_index, ps->m);
            g_write_block_if_ready(ps, w, data, len, base_index, finish);
            if (m2.length != 0 && m2.start < ps->last_match) {
                m2.length = (uint16_t)(m2.length - (ps->last_match - m2.start));
                m2.start = ps->last_match;
                if (m2.length < 4) {
                    m2 = gm_empty();
                }
            }
        }
        ps->m = m2;
    }
    return g_end_compress(ps, w, data, len, base_index, start, finish);
}

/* RleParser::compress (parse/rle.rs:22-47) */
static size_t g_rle_compress(GParser *ps, GBitWriter *w, const uint8_t *data, size_t len, uint32_t base_index,
                             size_t start, int finish) {
    g_start_compress(ps, base_index, start);
    size_t lookahead = finish ? 7 : 258;
    size_t max_ip = len > lookahead ? len - lookahead : 0;
    for (;;) {
        GMatch m = g_advance_to_match(ps, data, len, base_index, max_ip);
        if (m.length == 0) {
            break;
        }
        ps->ip = gm_end(m);
        g_insert_match(ps, base_index, m);
        g_write_block_if_ready(ps, w, data, len, base_index, finish);
    }
    return g_end_compress(ps, w, data, len, base_index, start, finish);
}

/* CompressorInner::compress (compress/mod.rs:226-290) for the Fast / Rle variants */
static size_t g_inner_compress(GParser *ps, GBitWriter *w, const uint8_t *data, size_t len, uint32_t base_index,
                               size_t start, int finish) {
    if (finish && len == start) { /* :234-238 */
        gbw_write_bits(w, 3, 10);
        gbw_flush(w);
        return 0;
    }
    return ps->use_hash ? g_greedy_compress(ps, w, data, len, base_index, start, finish)
                        : g_rle_compress(ps, w, data, len, base_index, start, finish);
}

/* compress_to_vec (level 1, compress/mod.rs:294-303) / compress_to_vec_rle (:306-310):
 * Compressor::new / new_rle (:69-123), ONE write_data (:126-190, the "no buffered input" branch:
 * the parser runs over the caller's buffer with Flush::None, then the window tail is kept), then
 * finish (:194-214) over the kept tail with Flush::Finish.  Returns bytes written, 0 if out_cap is
 * too small. */
static size_t g_compress(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap, int rle) {
    ensure_tables();
    if (len > (1u << 30)) {
        return 0; /* write_data splits inputs above 1 GiB into several calls (:130-136): not restated */
    }
    GBitWriter w = {0, 0, out, out_cap, 0, 0};
    const uint8_t hdr[2] = {0x78, 0x01};
    gbw_raw(&w, hdr, 2);
    GParser ps;
    memset(&ps, 0, sizeof(ps));
    ps.use_hash = !rle;
    ps.skip_ahead_shift = 5; /* GreedyParser::new(5, ..) :76 / RleParser::new(5) :114 */
    ps.hash_table = rle ? NULL : (uint32_t *)calloc(65536, sizeof(uint32_t));
    ps.symbols = (GSymbol *)malloc(sizeof(GSymbol) * G_MAX_SYMBOLS);
    size_t window_size = rle ? 1 : 32768;
    uint32_t adler = fdo_adler32(input, len);
    /* write_data */
    size_t written = g_inner_compress(&ps, &w, input, len, 0, 0, 0);
    size_t start = written > window_size ? written - window_size : 0;
    const uint8_t *kept = input + start; /* input.data = data[start..] */
    size_t kept_len = len - start;
    uint32_t base_index = (uint32_t)start;
    size_t input_written = written - start;
    /* finish */
    g_inner_compress(&ps, &w, kept, kept_len, base_index, input_written, 1);
    gbw_flush(&w);
    const uint8_t tr[4] = {(uint8_t)(adler >> 24), (uint8_t)(adler >> 16), (uint8_t)(adler >> 8), (uint8_t)adler};
    gbw_raw(&w, tr, 4);
    free(ps.hash_table);
    free(ps.symbols);
    return w.overflow ? 0 : w.pos;
}

size_t fdo_compress_bound(size_t len) { return len + len / 2 + 1024; }
size_t fdo_compress_level1(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap) {
    return g_compress(input, len, out, out_cap, 0);
}
size_t fdo_compress_rle(const uint8_t *input, size_t len, uint8_t *out, size_t out_cap) {
    return g_compress(input, len, out, out_cap, 1);
}

/* ========================================================================= */
/* PNG scanline filters (the step either side of the codec in the PNG pipeline) */
/* ========================================================================= */
/* Not part of the fdeflate crate: the `png` crate (image-rs/image-png, a reverse dependency;
 * README.md:11 of the reference) filters scanlines before compress_to_vec_ultra_fast and
 * reconstructs them after Decompressor::read.  That crate is not in /root/reference, so this is a
 * restatement of the published algorithm: PNG specification (W3C / ISO/IEC 15948), section 9.2
 * "Filter types for filter method 0" and 9.4 "Filter type 4: Paeth" -- filter types 0 None, 1 Sub,
 * 2 Up, 3 Average, 4 Paeth over bytes, `bpp` = bytes per complete pixel (>= 1), unsigned arithmetic
 * modulo 256, bytes to the left of the first pixel and above the first row are zero.
 * Parity pin: the specification's own definitions (there is no reference code or vector in the
 * tree for this row); tests round-trip filter -> unfilter and check hand-computed cases. */
static uint8_t png_paeth(uint8_t a, uint8_t b, uint8_t c) {
    int p = (int)a + (int)b - (int)c;
    int pa = p > a ? p - a : a - p;
    int pb = p > b ? p - b : b - p;
    int pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) {
        return a;
    }
    return pb <= pc ? b : c;
}

/* filt: rows x (1 + row_bytes) bytes (filter-type byte first); pix: rows x row_bytes.
 * Returns 0, or 1 if a filter type is > 4, or 2 if len is not a whole number of rows. */
int fdo_png_unfilter(const uint8_t *filt, size_t len, size_t row_bytes, size_t bpp, uint8_t *pix) {
    if (row_bytes == 0 || bpp == 0 || len % (row_bytes + 1) != 0) {
        return 2;
    }
    size_t rows = len / (row_bytes + 1);
    for (size_t r = 0; r < rows; r++) {
        const uint8_t *f = filt + r * (row_bytes + 1);
        uint8_t *cur = pix + r * row_bytes;
        const uint8_t *up = r ? cur - row_bytes : NULL;
        uint8_t t = f[0];
        if (t > 4) {
            return 1;
        }
        for (size_t x = 0; x < row_bytes; x++) {
            uint8_t a = x >= bpp ? cur[x - bpp] : 0;
            uint8_t b = up ? up[x] : 0;
            uint8_t c = (up && x >= bpp) ? up[x - bpp] : 0;
            uint8_t pred = t == 0 ? 0 : t == 1 ? a : t == 2 ? b : t == 3 ? (uint8_t)(((unsigned)a + b) >> 1) : png_paeth(a, b, c);
            cur[x] = (uint8_t)(f[1 + x] + pred);
        }
    }
    return 0;
}

/* pix: rows x row_bytes; types[rows]; filt: rows x (1 + row_bytes). */
int fdo_png_filter(const uint8_t *pix, size_t len, size_t row_bytes, size_t bpp, const uint8_t *types, uint8_t *filt) {
    if (row_bytes == 0 || bpp == 0 || len % row_bytes != 0) {
        return 2;
    }
    size_t rows = len / row_bytes;
    for (size_t r = 0; r < rows; r++) {
        const uint8_t *cur = pix + r * row_bytes;
        const uint8_t *up = r ? cur - row_bytes : NULL;
        uint8_t *f = filt + r * (row_bytes + 1);
        uint8_t t = types[r];
        if (t > 4) {
            return 1;
        }
        f[0] = t;
        for (size_t x = 0; x < row_bytes; x++) {
            uint8_t a = x >= bpp ? cur[x - bpp] : 0;
            uint8_t b = up ? up[x] : 0;
            uint8_t c = (up && x >= bpp) ? up[x - bpp] : 0;
            uint8_t pred = t == 0 ? 0 : t == 1 ? a : t == 2 ? b : t == 3 ? (uint8_t)(((unsigned)a + b) >> 1) : png_paeth(a, b, c);
            f[1 + x] = (uint8_t)(cur[x] - pred);
        }
    }
    return 0;
}
