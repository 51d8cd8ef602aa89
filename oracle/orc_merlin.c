/* oracle/orc_merlin.c -- TEST INFRASTRUCTURE ONLY.  See orc_merlin.h. */
#include "orc_merlin.h"
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
static const int RHO[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
static const int PI[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };

static inline uint64_t rol(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

void orc_keccak_f1600(uint8_t st[200]) {
    uint64_t a[25], bc[5];
    for (int i = 0; i < 25; i++) { uint64_t v = 0; for (int j = 7; j >= 0; j--) v = (v << 8) | st[8 * i + j]; a[i] = v; }
    for (int r = 0; r < 24; r++) {
        for (int i = 0; i < 5; i++) bc[i] = a[i] ^ a[i + 5] ^ a[i + 10] ^ a[i + 15] ^ a[i + 20];
        for (int i = 0; i < 5; i++) { uint64_t t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) a[j + i] ^= t; }
        uint64_t t = a[1];
        for (int i = 0; i < 24; i++) { int j = PI[i]; uint64_t b = a[j]; a[j] = rol(t, RHO[i]); t = b; }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = a[j + i];
            for (int i = 0; i < 5; i++) a[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        a[0] ^= RC[r];
    }
    for (int i = 0; i < 25; i++) for (int j = 0; j < 8; j++) st[8 * i + j] = (uint8_t)(a[i] >> (8 * j));
}

void orc_shake256(const uint8_t* msg, size_t len, uint8_t* out, size_t out_len) {
    enum { RATE = 136 };
    uint8_t st[200] = {0};
    size_t pos = 0;
    for (size_t i = 0; i < len; i++) { st[pos++] ^= msg[i]; if (pos == RATE) { orc_keccak_f1600(st); pos = 0; } }
    st[pos] ^= 0x1f; st[RATE - 1] ^= 0x80;
    orc_keccak_f1600(st);
    pos = 0;
    for (size_t i = 0; i < out_len; i++) { if (pos == RATE) { orc_keccak_f1600(st); pos = 0; } out[i] = st[pos++]; }
}

enum { STROBE_R = 166, FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

static void run_f(orc_transcript* t) {
    t->st[t->pos] ^= t->pos_begin;
    t->st[t->pos + 1] ^= 0x04;
    t->st[STROBE_R + 1] ^= 0x80;
    orc_keccak_f1600(t->st);
    t->pos = 0; t->pos_begin = 0;
}
static void absorb(orc_transcript* t, const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { t->st[t->pos++] ^= d[i]; if (t->pos == STROBE_R) run_f(t); }
}
static void squeeze(orc_transcript* t, uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { d[i] = t->st[t->pos]; t->st[t->pos++] = 0; if (t->pos == STROBE_R) run_f(t); }
}
static void begin_op(orc_transcript* t, uint8_t flags, int more) {
    if (more) return;   /* continuation of the current operation (caller guarantees same flags) */
    uint8_t hdr[2] = { t->pos_begin, flags };
    t->pos_begin = (uint8_t)(t->pos + 1);
    t->cur_flags = flags;
    absorb(t, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && t->pos != 0) run_f(t);
}
static void meta_ad(orc_transcript* t, const uint8_t* d, size_t n, int more) { begin_op(t, FLAG_M | FLAG_A, more); absorb(t, d, n); }
static void ad(orc_transcript* t, const uint8_t* d, size_t n, int more) { begin_op(t, FLAG_A, more); absorb(t, d, n); }
static void prf(orc_transcript* t, uint8_t* d, size_t n, int more) { begin_op(t, FLAG_I | FLAG_A | FLAG_C, more); squeeze(t, d, n); }

void orc_transcript_init(orc_transcript* t, const uint8_t* label, size_t label_len) {
    memset(t, 0, sizeof *t);
    const uint8_t hdr[6] = { 1, STROBE_R + 2, 1, 0, 1, 96 };
    memcpy(t->st, hdr, 6);
    memcpy(t->st + 6, "STROBEv1.0.2", 12);
    orc_keccak_f1600(t->st);
    meta_ad(t, (const uint8_t*)"Merlin v1.0", 11, 0);
    orc_transcript_append(t, (const uint8_t*)"dom-sep", 7, label, label_len);
}

void orc_transcript_append(orc_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    uint8_t len4[4] = { (uint8_t)msg_len, (uint8_t)(msg_len >> 8), (uint8_t)(msg_len >> 16), (uint8_t)(msg_len >> 24) };
    meta_ad(t, label, label_len, 0);
    meta_ad(t, len4, 4, 1);
    ad(t, msg, msg_len, 0);
}

void orc_transcript_append_u64(orc_transcript* t, const uint8_t* label, size_t label_len, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    orc_transcript_append(t, label, label_len, b, 8);
}

void orc_transcript_challenge(orc_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    uint8_t len4[4] = { (uint8_t)out_len, (uint8_t)(out_len >> 8), (uint8_t)(out_len >> 16), (uint8_t)(out_len >> 24) };
    meta_ad(t, label, label_len, 0);
    meta_ad(t, len4, 4, 1);
    prf(t, out, out_len, 0);
}
