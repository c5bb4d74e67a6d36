/* TEST INFRASTRUCTURE - CPU restatement of CAGRA's graph::optimize. Only tests/ may call this; the product path
 * (cuvs_amd/csrc/cagra.hip, cagra_mst.hip) never does.
 *
 * Follows cpp/src/neighbors/detail/cagra/graph_core.cuh of the reference:
 *   prune        kern_fused_prune :206-330 - 2-hop detour counts by rank (first rank kAB > kAD that lists B), self edges
 *                start at K, counts saturate at 0xffff, then `degree` selections of the smallest (count, rank) with every
 *                copy of the selected id retired
 *   reverse      kern_make_rev_graph :178-200 - the first `degree` reverse edges of a node; the reference fills them with
 *                racing atomicAdd tickets rank by rank, this library defines the order inside a rank as source-ascending
 *   merge        kern_merge_graph :375-470 - protected head = spanning-forest edges + pruned edges not among them, at
 *                least degree / 2 entries; reverse edges inserted behind the head, last first
 *   connectivity mst_optimization :1186-1581 - rounds over the edge rank k; a node with outgoing slots left whose rank-k
 *                neighbour lies in another component asks for an edge to it, or to the first incoming neighbour of it
 *                that has an incoming slot left (kern_mst_opt_update_graph :487-574); components are relabelled after
 *                every round (:577-613); full outgoing budgets grow by one while the row has room (:693-699); the last
 *                round walks i + 97 m towards the largest component (:1288-1316); rows de-duplicated at the end
 *                (:1551-1576). The reference lets thread timing decide which of several requests for one target is
 *                granted; this library grants the smallest requesting node id per target and round, adds a mutual pair
 *                once, lets only component roots ask in the first pass of the last round, and repeats the last round
 *                while it still joins components - so the forest is a function of the kNN graph. PARITY UNPINNED against
 *                the reference for this pass (its result is timing dependent): tests check the properties it guarantees
 *                (one component, protected edges present) and GPU == this restatement.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NONE 0xffffffffu

static void prune_row(const uint32_t* knn, int64_t n, uint32_t K, uint32_t degree, int64_t nid, uint32_t* det, uint32_t* out)
{
  const uint32_t* row = knn + nid * K;
  for (uint32_t k = 0; k < K; ++k) det[k] = row[k] == (uint32_t)nid ? K : 0;
  for (uint32_t kAD = 0; kAD + 1 < K; ++kAD) {
    const uint32_t iD = row[kAD];
    if (iD >= n) continue;
    for (uint32_t kDB = 0; kDB < K; ++kDB) {
      const uint32_t cand = knn[(int64_t)iD * K + kDB];
      for (uint32_t kAB = kAD + 1; kAB < K; ++kAB)
        if (row[kAB] == cand) { det[kAB] += 1; break; }
    }
  }
  for (uint32_t k = 0; k < K; ++k) {
    if (det[k] > 0xffffu) det[k] = 0xffffu;
    if (row[k] >= n) det[k] = 0xffffu;
  }
  for (uint32_t i = 0; i < degree; ++i) {
    uint32_t best = NONE;
    for (uint32_t k = 0; k < K; ++k) {
      const uint32_t tag = (det[k] << 16) | k;
      if (det[k] < 0xffffu && tag < best) best = tag;
    }
    uint32_t sel = NONE;
    if (best != NONE) {
      sel = row[best & 0xffffu];
      for (uint32_t k = 0; k < K; ++k)
        if (row[k] == sel) det[k] = 0xffffu;
    }
    out[nid * degree + i] = sel;
  }
}

static uint32_t root_of(const uint32_t* label, uint32_t x)
{
  while (label[x] != x) x = label[x];
  return x;
}

/* spanning forest: mst [n, degree] front-packed rows, cnt [n]; returns the number of components left */
int64_t oracle_cagra_mst(const uint32_t* knn, int64_t n, uint32_t K, uint32_t degree, uint32_t* mst, uint32_t* cnt)
{
  uint32_t* label   = malloc(sizeof(uint32_t) * n);
  uint32_t* out_cnt = calloc(n, sizeof(uint32_t));
  uint32_t* in_cnt  = calloc(n, sizeof(uint32_t));
  uint32_t* out_max = malloc(sizeof(uint32_t) * n);
  uint32_t* prop    = malloc(sizeof(uint32_t) * n);
  uint32_t* win     = malloc(sizeof(uint32_t) * n);
  uint32_t* size    = malloc(sizeof(uint32_t) * n);
  for (int64_t i = 0; i < n; ++i) {
    label[i]   = (uint32_t)i;
    out_max[i] = degree < 2 ? degree : 2;
    win[i]     = NONE;
    for (uint32_t k = 0; k < degree; ++k) mst[i * degree + k] = NONE;
  }
  int64_t clusters = n;
  int last_rounds  = 0;
  for (uint32_t k = 0; k <= K && clusters > 1;) {
    uint32_t main_label = NONE;
    if (k == K) {
      memset(size, 0, sizeof(uint32_t) * n);
      for (int64_t i = 0; i < n; ++i) size[label[i]] += 1;
      uint32_t best = 0;
      for (int64_t i = 0; i < n; ++i)
        if (size[i] > best) { best = size[i]; main_label = (uint32_t)i; }
    }
    /* propose (all reads are round-start state) */
    for (int64_t i = 0; i < n; ++i) {
      uint32_t t = NONE;
      const int asks = out_cnt[i] < out_max[i] && !(k == K && last_rounds == 0 && label[i] != (uint32_t)i);
      if (asks) {
        const uint32_t li = label[i];
        uint32_t j        = NONE;
        if (k < K) {
          j = knn[i * K + k];
        } else if (li != main_label) {
          int64_t w = (i + (int64_t)97 * (1 + last_rounds)) % n;
          /* bounded as in cagra_mst.hip: one turn of stride 97, then node by node */
          for (int64_t st = 0; st <= n / 97 && label[w] != main_label; ++st) w = (w + 97) % n;
          for (int64_t st = 0; st < n && label[w] != main_label; ++st) w = (w + 1) % n;
          j = label[w] == main_label ? (uint32_t)w : NONE;
        }
        if (j < (uint32_t)n && label[j] != li) {
          if (in_cnt[j] < degree - out_max[j]) {
            t = j;
          } else {
            for (uint32_t kj = 0; kj < degree; ++kj) {
              const uint32_t l = mst[((int64_t)j + 1) * degree - 1 - kj];
              if (l >= (uint32_t)n) continue;
              if (in_cnt[l] >= degree - out_max[l]) continue;
              t = l;
              break;
            }
          }
        }
      }
      prop[i] = t;
      if (t != NONE && (uint32_t)i < win[t]) win[t] = (uint32_t)i;
    }
    /* accept + hook */
    int64_t accepted = 0;
    for (int64_t i = 0; i < n; ++i) {
      const uint32_t t = prop[i];
      if (t == NONE || win[t] != (uint32_t)i) continue;
      if (prop[t] == (uint32_t)i && win[i] == t && t < (uint32_t)i) continue; /* mutual pair: the smaller id adds it */
      mst[i * degree + out_cnt[i]++]                   = t;
      mst[((int64_t)t + 1) * degree - 1 - in_cnt[t]++] = (uint32_t)i;
      ++accepted;
      const uint32_t ra = root_of(label, (uint32_t)i), rb = root_of(label, t);
      if (ra != rb) label[ra > rb ? ra : rb] = ra < rb ? ra : rb;
    }
    clusters = 0;
    for (int64_t i = 0; i < n; ++i) {
      label[i] = root_of(label, (uint32_t)i); /* ascending i: label[root] is final before its members are visited */
      win[i]   = NONE;
      if (out_cnt[i] == out_max[i] && out_cnt[i] + in_cnt[i] < degree) out_max[i] += 1;
      clusters += label[i] == (uint32_t)i;
    }
    if (k < K) {
      ++k;
    } else {
      ++last_rounds;
      if ((accepted == 0 && last_rounds > 1) || last_rounds >= 64) break;
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    uint32_t* row = mst + i * degree;
    uint32_t c    = 0;
    for (uint32_t kj = 0; kj < degree; ++kj) {
      const uint32_t j = row[kj];
      if (j >= (uint32_t)n) continue;
      int dup = 0;
      for (uint32_t ki = 0; ki < c; ++ki) dup |= row[ki] == j;
      if (!dup) row[c++] = j;
    }
    cnt[i] = c;
    for (uint32_t kj = c; kj < degree; ++kj) row[kj] = NONE;
  }
  free(label); free(out_cnt); free(in_cnt); free(out_max); free(prop); free(win); free(size);
  return clusters;
}

/* knn [n, K] -> out [n, degree]; returns the components left by the connectivity pass (0 when it was not asked for) */
int64_t oracle_cagra_optimize(const uint32_t* knn, int64_t n, uint32_t K, uint32_t degree, int guarantee_connectivity,
                              uint32_t* out)
{
  uint32_t* det = malloc(sizeof(uint32_t) * K);
  for (int64_t nid = 0; nid < n; ++nid) prune_row(knn, n, K, degree, nid, det, out);
  free(det);
  /* reverse edges in (rank, source) order, the first `degree` per node */
  uint32_t* rev     = malloc(sizeof(uint32_t) * n * degree);
  uint32_t* rev_cnt = calloc(n, sizeof(uint32_t));
  for (uint32_t k = 0; k < degree; ++k)
    for (int64_t src = 0; src < n; ++src) {
      const uint32_t d = out[src * degree + k];
      if (d >= (uint32_t)n) continue;
      if (rev_cnt[d] < degree) rev[(int64_t)d * degree + rev_cnt[d]++] = (uint32_t)src;
    }
  uint32_t *mst = NULL, *mst_cnt = NULL;
  int64_t left  = 0;
  if (guarantee_connectivity) {
    mst     = malloc(sizeof(uint32_t) * n * degree);
    mst_cnt = malloc(sizeof(uint32_t) * n);
    left    = oracle_cagra_mst(knn, n, K, degree, mst, mst_cnt);
  }
  uint32_t* row = malloc(sizeof(uint32_t) * degree);
  for (int64_t nid = 0; nid < n; ++nid) {
    uint32_t n_mst = 0;
    if (mst) {
      n_mst = mst_cnt[nid] < degree ? mst_cnt[nid] : degree;
      for (uint32_t i = 0; i < n_mst; ++i) row[i] = mst[nid * degree + i];
      uint32_t o = n_mst;
      for (uint32_t pj = 0; pj < degree && o < degree; ++pj) {
        const uint32_t v = out[nid * degree + pj];
        int dup = 0;
        for (uint32_t m = 0; m < o; ++m) dup |= row[m] == v;
        if (!dup) row[o++] = v;
      }
      for (; o < degree; ++o) row[o] = NONE;
    } else {
      memcpy(row, out + nid * degree, sizeof(uint32_t) * degree);
    }
    const uint32_t prot = n_mst > degree / 2 ? n_mst : degree / 2;
    if (prot < degree) {
      uint32_t kr = rev_cnt[nid] < degree ? rev_cnt[nid] : degree;
      while (kr) {
        kr -= 1;
        const uint32_t rv = rev[nid * degree + kr];
        uint32_t pos      = degree;
        for (uint32_t i = 0; i < degree; ++i)
          if (row[i] == rv) { pos = i; break; }
        if (pos < prot) continue;
        uint32_t shift = pos - prot;
        if (pos >= degree) shift = degree - prot - 1;
        memmove(row + prot + 1, row + prot, sizeof(uint32_t) * shift);
        row[prot] = rv;
      }
    }
    memcpy(out + nid * degree, row, sizeof(uint32_t) * degree);
  }
  free(row); free(rev); free(rev_cnt); free(mst); free(mst_cnt);
  return left;
}
