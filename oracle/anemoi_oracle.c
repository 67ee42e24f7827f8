/*
 * anemoi_oracle.c -- CPU restatement of the reference's Anemoi / Jive / Sponge path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP kernels are compared against and the
 * "port" CPU baseline timed by bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; nothing under anemoi-rust_amd/ links, includes or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against every known-answer
 * vector the reference's own #[test] functions hold (tests/golden/kats.json: 14 instances x
 * {10 sbox_layer, 10 hash_field, 4 hash(bytes), 4 compress/compress_k(2)/merge} + 7 x 4
 * compress_k(4)), and against the independent Python big-int restatement oracle/anemoi_ref.py.
 * The reference itself (Rust + un-vendored arkworks crates, no Cargo.lock) cannot be built in
 * this image (no rustc/cargo), so there is no oracle/_ref build; see DESIGN.md.
 *
 * Field arithmetic: the reference delegates to arkworks ark-ff ^0.4 `Fp<MontBackend<_,N>,N>`
 * (Cargo.toml:15-22; not vendored).  Restated here from its published algorithm: N little-endian
 * u64 limbs, Montgomery form with R = 2^(64N), values kept fully reduced, CIOS multiplication.
 * Element encoding at this file's API = that in-memory form (what a Rust `&[Felt]` holds).
 *
 * Reference lines followed (paths relative to the reference crate root):
 *   orc_mul_by_g       src/traits.rs:78-91
 *   orc_ark_layer      src/traits.rs:111-125
 *   orc_mds_layer      src/traits.rs:136-157   (arms 1 and 2; no shipped instance has more columns)
 *   orc_sbox_layer     src/traits.rs:326-358
 *   orc_permutation    src/traits.rs:361-378
 *   orc_exp_inv_alpha  src/<f>/sbox.rs `exp_by_inv_alpha`: x^INV_ALPHA.  The reference hard-codes an
 *                      addition chain (454 ops for bls12_381); any chain yields the same canonical
 *                      value, so a 5-bit sliding window (~460 ops) is used here.
 *   orc_compress_k     src/<f>/anemoi_2_1/hasher.rs:96-110, anemoi_4_3/hasher.rs:148-179
 *   orc_hash_field     src/<f>/anemoi_2_1/hasher.rs:68-85,  anemoi_4_3/hasher.rs:93-129
 *   orc_hash_bytes     src/<f>/anemoi_2_1/hasher.rs:18-66,  anemoi_4_3/hasher.rs:19-91
 *   orc_merge          src/<f>/anemoi_2_1/hasher.rs:87-92,  anemoi_4_3/hasher.rs:131-145
 *   orc_digest_bytes   src/<f>/anemoi_x/digest.rs:42-46 (serialize_compressed = LE canonical bytes)
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "anemoi_params_gen.h"

typedef unsigned __int128 u128;
#define INL static inline __attribute__((always_inline))

/* ------------------------------------------------------------------ field arithmetic */

INL int fe_geq(const uint64_t *a, const uint64_t *b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return 1;
}

INL uint64_t fe_sub_raw(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
  uint64_t borrow = 0;
  for (int i = 0; i < n; i++) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}

INL void fe_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const orc_field *F, int n) {
  uint64_t carry = 0, t[ORC_MAXL];
  for (int i = 0; i < n; i++) {
    u128 s = (u128)a[i] + b[i] + carry;
    t[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  if (carry || fe_geq(t, F->p, n)) fe_sub_raw(t, t, F->p, n);
  for (int i = 0; i < n; i++) r[i] = t[i];
}

INL void fe_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const orc_field *F, int n) {
  uint64_t t[ORC_MAXL];
  if (fe_sub_raw(t, a, b, n)) {
    uint64_t carry = 0;
    for (int i = 0; i < n; i++) {
      u128 s = (u128)t[i] + F->p[i] + carry;
      t[i] = (uint64_t)s;
      carry = (uint64_t)(s >> 64);
    }
  }
  for (int i = 0; i < n; i++) r[i] = t[i];
}

/* CIOS Montgomery product a*b/R mod p, inputs and output fully reduced. */
INL void fe_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const orc_field *F, int n) {
  uint64_t t[ORC_MAXL + 2];
  for (int i = 0; i < n + 2; i++) t[i] = 0;
  _Pragma("GCC unroll 8")
  for (int i = 0; i < n; i++) {
    uint64_t c = 0;
    _Pragma("GCC unroll 8")
    for (int j = 0; j < n; j++) {
      u128 s = (u128)a[j] * b[i] + t[j] + c;
      t[j] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    u128 s = (u128)t[n] + c;
    t[n] = (uint64_t)s;
    t[n + 1] = (uint64_t)(s >> 64);
    uint64_t m = t[0] * F->n0inv;
    s = (u128)m * F->p[0] + t[0];
    c = (uint64_t)(s >> 64);
    _Pragma("GCC unroll 8")
    for (int j = 1; j < n; j++) {
      s = (u128)m * F->p[j] + t[j] + c;
      t[j - 1] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    s = (u128)t[n] + c;
    t[n - 1] = (uint64_t)s;
    t[n] = t[n + 1] + (uint64_t)(s >> 64);
  }
  if (t[n] || fe_geq(t, F->p, n)) fe_sub_raw(t, t, F->p, n);
  for (int i = 0; i < n; i++) r[i] = t[i];
}

INL void fe_copy(uint64_t *r, const uint64_t *a, int n) {
  for (int i = 0; i < n; i++) r[i] = a[i];
}

/* src/traits.rs:78-91 -- doubling chains for g in {2,3,5,7,..}, a real product otherwise (g = 22). */
INL void orc_mul_by_g(uint64_t *r, const uint64_t *x, const orc_field *F, int n) {
  uint64_t d[ORC_MAXL], q[ORC_MAXL];
  switch (F->g) {
    case 2: fe_add(r, x, x, F, n); break;
    case 3: fe_add(d, x, x, F, n); fe_add(r, d, x, F, n); break;
    case 5: fe_add(d, x, x, F, n); fe_add(d, d, d, F, n); fe_add(r, d, x, F, n); break;
    case 7: fe_add(d, x, x, F, n); fe_add(d, d, x, F, n); fe_add(d, d, d, F, n); fe_add(r, d, x, F, n); break;
    case 15:
      fe_add(d, x, x, F, n); fe_add(d, d, d, F, n); fe_add(d, d, d, F, n); fe_add(d, d, d, F, n);
      fe_sub(r, d, x, F, n);
      break;
    default: fe_copy(q, x, n); fe_mul(r, F->gmont, q, F, n); break;
  }
}

/* x^INV_ALPHA by a 5-bit sliding window over odd powers. */
INL void orc_exp_inv_alpha(uint64_t *r, const uint64_t *x, const orc_field *F, int n) {
  uint64_t tab[16][ORC_MAXL], x2[ORC_MAXL], acc[ORC_MAXL] = {0, 0, 0, 0, 0, 0};
  fe_copy(tab[0], x, n);
  fe_mul(x2, x, x, F, n);
  for (int i = 1; i < 16; i++) fe_mul(tab[i], tab[i - 1], x2, F, n);
  int top = 64 * n - 1;
  while (top >= 0 && !((F->inv_alpha[top >> 6] >> (top & 63)) & 1)) top--;
  int started = 0;
  for (int i = top; i >= 0;) {
    if (!((F->inv_alpha[i >> 6] >> (i & 63)) & 1)) {
      fe_mul(acc, acc, acc, F, n);
      i--;
      continue;
    }
    int lo = i - 4 < 0 ? 0 : i - 4;
    while (!((F->inv_alpha[lo >> 6] >> (lo & 63)) & 1)) lo++;
    unsigned v = 0;
    for (int b = i; b >= lo; b--) v = (v << 1) | ((F->inv_alpha[b >> 6] >> (b & 63)) & 1);
    if (!started) {
      fe_copy(acc, tab[v >> 1], n);
      started = 1;
    } else {
      for (int b = i; b >= lo; b--) fe_mul(acc, acc, acc, F, n);
      fe_mul(acc, acc, tab[v >> 1], F, n);
    }
    i = lo - 1;
  }
  fe_copy(r, acc, n);
}

/* src/traits.rs:111-125 */
INL void orc_ark_layer(uint64_t *st, int r, const orc_field *F, int n, int cols) {
  const uint64_t *C = cols == 1 ? F->c21 : F->c43, *D = cols == 1 ? F->d21 : F->d43;
  for (int i = 0; i < cols; i++) {
    fe_add(st + i * n, st + i * n, C + (r * cols + i) * n, F, n);
    fe_add(st + (cols + i) * n, st + (cols + i) * n, D + (r * cols + i) * n, F, n);
  }
}

/* src/traits.rs:136-157 */
INL void orc_mds_layer(uint64_t *st, const orc_field *F, int n, int cols) {
  uint64_t t[ORC_MAXL];
  if (cols == 1) {
    fe_add(st + n, st + n, st, F, n);
    fe_add(st, st, st + n, F, n);
    return;
  }
  uint64_t *s0 = st, *s1 = st + n, *s2 = st + 2 * n, *s3 = st + 3 * n;
  orc_mul_by_g(t, s1, F, n); fe_add(s0, s0, t, F, n);
  orc_mul_by_g(t, s0, F, n); fe_add(s1, s1, t, F, n);
  orc_mul_by_g(t, s2, F, n); fe_add(s3, s3, t, F, n);
  orc_mul_by_g(t, s3, F, n); fe_add(s2, s2, t, F, n);
  fe_copy(t, s2, n); fe_copy(s2, s3, n); fe_copy(s3, t, n); /* swap(2,3) */
  fe_add(s2, s2, s0, F, n);
  fe_add(s3, s3, s1, F, n);
  fe_add(s0, s0, s2, F, n);
  fe_add(s1, s1, s3, F, n);
}

/* src/traits.rs:326-358 */
INL void orc_sbox_layer(uint64_t *st, const orc_field *F, int n, int cols) {
  uint64_t t[ORC_MAXL], u[ORC_MAXL];
  for (int i = 0; i < cols; i++) {
    uint64_t *x = st + i * n, *y = st + (cols + i) * n;
    fe_mul(t, y, y, F, n);
    orc_mul_by_g(u, t, F, n);
    fe_sub(x, x, u, F, n);              /* x -= g*y^2            */
    orc_exp_inv_alpha(t, x, F, n);
    fe_sub(y, y, t, F, n);              /* y -= x^(1/alpha)      */
    fe_mul(t, y, y, F, n);
    orc_mul_by_g(u, t, F, n);
    fe_add(x, x, u, F, n);
    fe_add(x, x, F->delta, F, n);       /* x += g*y^2 + delta    */
  }
}

/* src/traits.rs:361-378 */
INL void orc_permutation_n(uint64_t *st, const orc_field *F, int n, int cols) {
  int rounds = cols == 1 ? F->rounds21 : F->rounds43;
  for (int r = 0; r < rounds; r++) {
    orc_ark_layer(st, r, F, n, cols);
    orc_mds_layer(st, F, n, cols);
    orc_sbox_layer(st, F, n, cols);
  }
  orc_mds_layer(st, F, n, cols);
}

/* one specialised copy per (limbs, columns) so the limb loops unroll */
static void perm_4_1(uint64_t *s, const orc_field *F) { orc_permutation_n(s, F, 4, 1); }
static void perm_4_2(uint64_t *s, const orc_field *F) { orc_permutation_n(s, F, 4, 2); }
static void perm_6_1(uint64_t *s, const orc_field *F) { orc_permutation_n(s, F, 6, 1); }
static void perm_6_2(uint64_t *s, const orc_field *F) { orc_permutation_n(s, F, 6, 2); }
static void sbox_4_1(uint64_t *s, const orc_field *F) { orc_sbox_layer(s, F, 4, 1); }
static void sbox_4_2(uint64_t *s, const orc_field *F) { orc_sbox_layer(s, F, 4, 2); }
static void sbox_6_1(uint64_t *s, const orc_field *F) { orc_sbox_layer(s, F, 6, 1); }
static void sbox_6_2(uint64_t *s, const orc_field *F) { orc_sbox_layer(s, F, 6, 2); }

static const orc_field *get_field(int field, int width) {
  if (field < 0 || field >= ORC_NFIELDS || (width != 2 && width != 4)) return NULL;
  return &ORC_FIELDS[field];
}

static void permute(uint64_t *st, const orc_field *F, int width) {
  if (F->limbs == 4) (width == 2 ? perm_4_1 : perm_4_2)(st, F);
  else (width == 2 ? perm_6_1 : perm_6_2)(st, F);
}

/* ------------------------------------------------------------------ public API (C, ctypes) */

int orc_num_fields(void) { return ORC_NFIELDS; }
const char *orc_field_name(int field) { return field >= 0 && field < ORC_NFIELDS ? ORC_FIELDS[field].name : NULL; }
int orc_field_limbs(int field) { return field >= 0 && field < ORC_NFIELDS ? ORC_FIELDS[field].limbs : -1; }

/* canonical LE limbs -> Montgomery form (value must be < p) and back */
int orc_to_mont(int field, const uint64_t *in, uint64_t *out, size_t count) {
  const orc_field *F = get_field(field, 2);
  if (!F) return -1;
  for (size_t i = 0; i < count; i++) {
    uint64_t a[ORC_MAXL] = {0, 0, 0, 0, 0, 0};
    fe_copy(a, in + i * F->limbs, F->limbs);
    if (F->limbs == 4) fe_mul(out + i * 4, a, F->r2, F, 4); else fe_mul(out + i * 6, a, F->r2, F, 6);
  }
  return 0;
}

int orc_from_mont(int field, const uint64_t *in, uint64_t *out, size_t count) {
  const orc_field *F = get_field(field, 2);
  if (!F) return -1;
  uint64_t one[ORC_MAXL] = {1, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < count; i++) {
    uint64_t a[ORC_MAXL] = {0, 0, 0, 0, 0, 0};
    fe_copy(a, in + i * F->limbs, F->limbs);
    if (F->limbs == 4) fe_mul(out + i * 4, a, one, F, 4); else fe_mul(out + i * 6, a, one, F, 6);
  }
  return 0;
}

int orc_permutation(int field, int width, uint64_t *state) {
  const orc_field *F = get_field(field, width);
  if (!F) return -1;
  permute(state, F, width);
  return 0;
}

int orc_sbox_layer_state(int field, int width, uint64_t *state) {
  const orc_field *F = get_field(field, width);
  if (!F) return -1;
  if (F->limbs == 4) (width == 2 ? sbox_4_1 : sbox_4_2)(state, F);
  else (width == 2 ? sbox_6_1 : sbox_6_2)(state, F);
  return 0;
}

/* Jive: out[i] = sum_{j<k} elems[i + c*j] + perm(elems)[i + c*j],  c = width/k. */
int orc_compress_k(int field, int width, const uint64_t *elems, uint64_t *out, int k) {
  const orc_field *F = get_field(field, width);
  if (!F) return -1;
  if (k <= 0 || width % k != 0 || k % 2 != 0) return -2; /* the reference's assert!s */
  int n = F->limbs, c = width / k;
  uint64_t st[4 * ORC_MAXL];
  memcpy(st, elems, sizeof(uint64_t) * n * width);
  permute(st, F, width);
  for (int i = 0; i < c; i++) {
    uint64_t acc[ORC_MAXL] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < k; j++) {
      fe_add(acc, acc, elems + (i + c * j) * n, F, n);
      fe_add(acc, acc, st + (i + c * j) * n, F, n);
    }
    fe_copy(out + i * n, acc, n);
  }
  return 0;
}

int orc_compress(int field, int width, const uint64_t *elems, uint64_t *out) {
  return orc_compress_k(field, width, elems, out, 2);
}

int orc_hash_field(int field, int width, const uint64_t *elems, size_t count, uint64_t *out) {
  const orc_field *F = get_field(field, width);
  if (!F) return -1;
  int n = F->limbs, rate = width - 1;
  uint64_t st[4 * ORC_MAXL];
  memset(st, 0, sizeof st);
  int i = 0;
  for (size_t e = 0; e < count; e++) {
    fe_add(st + i * n, st + i * n, elems + e * n, F, n);
    if (++i == rate) {
      permute(st, F, width);
      i = 0;
    }
  }
  int sigma = count % rate == 0;
  if (sigma) {
    fe_add(st + (width - 1) * n, st + (width - 1) * n, F->one, F, n);
  } else {
    fe_add(st + i * n, st + i * n, F->one, F, n);
    permute(st, F, width);
  }
  fe_copy(out, st, n);
  return 0;
}

/* 31/47-byte LE chunk (zero-extended; 0x01 appended to a SHORT last chunk) -> Montgomery element.
 * from_le_bytes_mod_order: the integer is < 2^(8*chunk+1) < p here, the loop is for generality. */
static void chunk_to_elem(const orc_field *F, const uint8_t *src, size_t len, uint64_t *out) {
  int n = F->limbs;
  uint8_t buf[8 * ORC_MAXL];
  memset(buf, 0, sizeof buf);
  memcpy(buf, src, len);
  if (len < (size_t)F->chunk) buf[len] = 1;
  uint64_t v[ORC_MAXL];
  for (int i = 0; i < n; i++) {
    v[i] = 0;
    for (int b = 7; b >= 0; b--) v[i] = (v[i] << 8) | buf[8 * i + b];
  }
  while (fe_geq(v, F->p, n)) fe_sub_raw(v, v, F->p, n);
  if (n == 4) fe_mul(out, v, F->r2, F, 4); else fe_mul(out, v, F->r2, F, 6);
}

int orc_hash_bytes(int field, int width, const uint8_t *bytes, size_t len, uint64_t *out) {
  const orc_field *F = get_field(field, width);
  if (!F) return -1;
  size_t ch = (size_t)F->chunk, count = (len + ch - 1) / ch;
  uint64_t *elems = (uint64_t *)malloc(sizeof(uint64_t) * F->limbs * (count ? count : 1));
  if (!elems) return -3;
  for (size_t e = 0; e < count; e++) {
    size_t off = e * ch, l = len - off < ch ? len - off : ch;
    chunk_to_elem(F, bytes + off, l, elems + e * F->limbs);
  }
  int rc = orc_hash_field(field, width, elems, count, out);
  free(elems);
  return rc;
}

int orc_merge(int field, int width, const uint64_t *left, const uint64_t *right, uint64_t *out) {
  const orc_field *F = get_field(field, width);
  if (!F) return -1;
  int n = F->limbs;
  uint64_t st[4 * ORC_MAXL];
  memset(st, 0, sizeof st);
  if (width == 2) {
    fe_copy(st, left, n);
    fe_copy(st + n, right, n);
    return orc_compress(field, width, st, out);
  }
  /* anemoi_4_3/hasher.rs:136-138 copies digests[0] into BOTH rate cells; kept as is. */
  fe_copy(st, left, n);
  fe_copy(st + n, left, n);
  permute(st, F, width);
  fe_copy(out, st, n);
  return 0;
}

/* digest.rs:42-46: canonical little-endian bytes, 8*limbs of them */
int orc_digest_bytes(int field, const uint64_t *digest, uint8_t *out) {
  const orc_field *F = get_field(field, 2);
  if (!F) return -1;
  uint64_t c[ORC_MAXL];
  orc_from_mont(field, digest, c, 1);
  for (int i = 0; i < F->limbs; i++)
    for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(c[i] >> (8 * b));
  return 0;
}

/* Binary Merkle tree over 2^depth leaf digests with the 2-1 instance's merge (= Jive compress). */
int orc_merkle_root(int field, const uint64_t *leaves, unsigned depth, uint64_t *root) {
  const orc_field *F = get_field(field, 2);
  if (!F || depth > 30) return -1;
  int n = F->limbs;
  size_t cnt = (size_t)1 << depth;
  uint64_t *lvl = (uint64_t *)malloc(sizeof(uint64_t) * n * cnt);
  if (!lvl) return -3;
  memcpy(lvl, leaves, sizeof(uint64_t) * n * cnt);
  for (; cnt > 1; cnt >>= 1)
    for (size_t i = 0; i < cnt / 2; i++) orc_compress(field, 2, lvl + 2 * i * n, lvl + i * n);
  fe_copy(root, lvl, n);
  free(lvl);
  return 0;
}

/* ------------------------------------------------------------------ threaded batches (CPU baseline) */

typedef struct {
  int field, width, kind, k;
  const uint8_t *in;
  uint64_t *out;
  size_t begin, end, in_stride, msg_len;
} orc_job;

static void *orc_worker(void *arg) {
  orc_job *j = (orc_job *)arg;
  const orc_field *F = &ORC_FIELDS[j->field];
  size_t n = (size_t)F->limbs;
  for (size_t i = j->begin; i < j->end; i++) {
    if (j->kind == 0)
      orc_compress_k(j->field, j->width, (const uint64_t *)j->in + i * j->width * n,
                     j->out + i * (j->width / j->k) * n, j->k);
    else if (j->kind == 1)
      orc_hash_bytes(j->field, j->width, j->in + i * j->msg_len, j->msg_len, j->out + i * n);
    else
      orc_hash_field(j->field, j->width, (const uint64_t *)j->in + i * j->msg_len * n, j->msg_len, j->out + i * n);
  }
  return NULL;
}

static int run_batch(orc_job proto, size_t count, int threads) {
  if (!get_field(proto.field, proto.width)) return -1;
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  pthread_t tid[256];
  orc_job jobs[256];
  for (int t = 0; t < threads; t++) {
    jobs[t] = proto;
    jobs[t].begin = count * (size_t)t / (size_t)threads;
    jobs[t].end = count * (size_t)(t + 1) / (size_t)threads;
    if (threads == 1) orc_worker(&jobs[t]);
    else if (pthread_create(&tid[t], NULL, orc_worker, &jobs[t])) return -4;
  }
  if (threads > 1)
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
  return 0;
}

int orc_compress_batch(int field, int width, int k, const uint64_t *in, uint64_t *out, size_t count, int threads) {
  if (k <= 0 || (width != 2 && width != 4) || width % k != 0 || k % 2 != 0) return -2;
  orc_job j = {field, width, 0, k, (const uint8_t *)in, out, 0, 0, 0, 0};
  return run_batch(j, count, threads);
}

int orc_hash_bytes_batch(int field, int width, const uint8_t *msgs, size_t msg_len, size_t count, uint64_t *out, int threads) {
  orc_job j = {field, width, 1, 2, msgs, out, 0, 0, 0, msg_len};
  return run_batch(j, count, threads);
}

int orc_hash_field_batch(int field, int width, const uint64_t *elems, size_t elems_per_msg, size_t count, uint64_t *out, int threads) {
  orc_job j = {field, width, 2, 2, (const uint8_t *)elems, out, 0, 0, 0, elems_per_msg};
  return run_batch(j, count, threads);
}
