/* tests/model/range_model.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU model of the arithmetic behind the entropy kernels' batched update (sqz_amd/csrc/sqz_tree.h,
 * bump_batch): up to 64 tokens update the adaptive Huffman tree in one step WITHOUT walking any
 * leaf->root chain.  This model keeps numeric intervals per node; the kernels keep, per node, its
 * first and last LEAF and per leaf its position (the same intervals, cheaper to repair: only
 * leaves are rewritten by a restructure) -- the tests n(v), the violation rule and the repairs
 * per sibling exchange / promotion / insert are the ones modelled here:
 *
 *   every leaf has a position in depth-first order (lo before hi), every node the interval
 *   [st, en) of the leaf positions below it.  Then for a batch of symbols
 *       n(v) = P[en(v)] - P[st(v)],   P = prefix sums of the batch's histogram over positions,
 *   is how many of the batch's chains pass through node v -- for ALL nodes at once, one
 *   independent lookup each.  The reference would change no link during the batch iff, for
 *   every node v the batch touches (parent p, sibling s, uncle u; f = counts before the batch):
 *       v is lo(p):                 f(v) + n(v) <= f(s)
 *       v is hi(p), p not the root: f(v) + n(v) <= f(u)
 *   (the proof is in the head comment of sqz_tree.h; the tests are the same, only
 *   n(v) comes from the intervals instead of counters along the chains).  If some v fails, the
 *   first token that may not be applied is the (f(bound) - f(v) + 1)-th one through v; the
 *   tokens in front of the earliest such token are applied, the token itself takes the exact
 *   one-at-a-time path (the reference's sequence), which also keeps the intervals, the
 *   per-leaf codes, the cached test partners and the depth mark up to date with FLAT passes
 *   over the node arrays (one pass per sibling exchange / promotion / insert, no subtree walk).
 *
 * This file restates that in plain C with every "for all nodes" loop written as the flat pass
 * the kernels run (one lane per node), drives it with symbol sequences and compares it in
 * lockstep with the oracle's tree (oracle/sqz_oracle.c, included below for its static tree
 * functions): same links, counts, depths, depth mark and codes after every step, same stats.
 * huffman.h line numbers are the reference's (attic/map_experiment).
 *
 * Build + run: tests/test_range_model.py (gcc, CPU only).
 */
#define _POSIX_C_SOURCE 200809L
#include "../../oracle/sqz_oracle.c"

#include <assert.h>

#define M_NIL (-1)
#define M_MAXN 640
#define M_BATCH 64

typedef struct {
    int leaves, nodes, ref_leaves;   /* ids: leaves [0, leaves), root = leaves, internals up from it */
    int next, idend;
    int up[M_MAXN], lo[M_MAXN], hi[M_MAXN];
    uint32_t freq[M_MAXN];
    int dep[M_MAXN];
    int st[M_MAXN], en[M_MAXN];      /* leaf-position interval; valid for attached nodes */
    int partner[M_MAXN];             /* node whose count bounds mine (sibling / uncle), -1 = untested */
    uint64_t code[M_MAXN];           /* leaves: code in stream order, dep[] bits */
    int mark;                        /* huffman.h:26 depth high-water mark */
    int complete;
    /* huffman.h:29-33 */
    uint64_t st_updates, st_swaps, st_moves;
} mtree;

static int m_attached(const mtree* t, int v) { return v == t->leaves || t->up[v] != M_NIL; }

static void m_init(mtree* t, int leaves, int nodes, int ref_leaves) {
    memset(t, 0, sizeof(*t));
    t->leaves = leaves; t->nodes = nodes; t->ref_leaves = ref_leaves;
    t->next = leaves + 1;
    t->idend = leaves + 1 + ref_leaves - 2 < nodes ? leaves + 1 + ref_leaves - 2 : nodes;
    for (int v = 0; v < nodes; v++) {
        t->up[v] = t->lo[v] = t->hi[v] = M_NIL; t->partner[v] = M_NIL;
        t->st[v] = t->en[v] = 0;
    }
}

/* ---- flat passes ------------------------------------------------------------------------ */

/* the test partner of every node inside [a, b) (or everywhere): sibling for a lo child, uncle
 * for a hi child below the root's children */
static void m_partners(mtree* t, int a, int b) {
    for (int v = 0; v < t->next; v++) {                       /* one lane per node */
        if (!m_attached(t, v)) { continue; }
        if (v == t->leaves) { t->partner[v] = M_NIL; continue; }
        if (t->st[v] < a || t->en[v] > b) { continue; }
        const int p = t->up[v];
        const int is_hi = t->hi[p] == v;
        if (!is_hi) { t->partner[v] = t->hi[p]; continue; }    /* may be nil while the root has one child */
        const int g = t->up[p];
        if (g == M_NIL) { t->partner[v] = M_NIL; continue; }
        t->partner[v] = t->lo[g] == p ? t->hi[g] : t->lo[g];
    }
}

static void m_mark_reset_if_root(mtree* t, int top) { if (top == t->leaves) { t->mark = 0; } }

/* huffman.h:41-62 as far as the mark and the statistics go: every node of top's subtree is
 * visited once; depths themselves are kept right by the passes below */
static void m_relabel_mark(mtree* t, int top) {
    m_mark_reset_if_root(t, top);
    const int a = t->st[top], b = t->en[top];
    for (int v = 0; v < t->next; v++) {                       /* one lane per node, max-reduce */
        if (!m_attached(t, v)) { continue; }
        if (t->st[v] >= a && t->en[v] <= b && (v == top || t->dep[v] > t->dep[top])) {
            t->st_updates++;
            if (t->dep[v] > t->mark) { t->mark = t->dep[v]; }
        }
    }
}

/* the two children of p have traded slots (links already say so): positions and codes follow.
 * X = child now in the lo slot (was hi: interval [m, b)), Y = child now hi (was [a, m)). */
static void m_swap_fix(mtree* t, int p) {
    const int X = t->lo[p], Y = t->hi[p];
    const int a = t->st[Y], m = t->st[X], b = t->en[X];
    assert(t->en[Y] == m && a == t->st[p] && b == t->en[p]);
    const int dp = t->dep[p];
    for (int v = 0; v < t->next; v++) {                       /* one lane per node */
        if (!m_attached(t, v) || v == p) { continue; }
        int shift = 0;
        if (t->st[v] >= m && t->en[v] <= b) { shift = -(m - a); }
        else if (t->st[v] >= a && t->en[v] <= m) { shift = b - m; }
        else { continue; }
        t->st[v] += shift; t->en[v] += shift;
        if (v < t->leaves) { t->code[v] ^= 1ull << (t->dep[v] - 1 - dp); }
    }
    /* X and Y changed sides: so did their tests; their children keep theirs (their uncle is
     * still the other one of the pair) */
    m_partners(t, a, b);
}

/* c (hi child of p) and its uncle u have traded places under g (links already say so):
 *   left  (p = lo(g)):  [x][c][u] -> [x][u][c]      c: G01S -> G1S     u: G1S -> G01S
 *   right (p = hi(g)):  [u][x][c] -> [c][x][u]      c: G11S -> G0S     u: G0S -> G11S
 * c's subtree comes up one level, u's goes down one. */
static void m_promote_fix(mtree* t, int g, int p, int c, int u, int left) {
    const int x = t->lo[p];
    assert(t->hi[p] == u);
    const int C = t->en[c] - t->st[c], U = t->en[u] - t->st[u], X = t->en[x] - t->st[x];
    const int ca = t->st[c], cb = t->en[c], ua = t->st[u], ub = t->en[u], xa = t->st[x], xb = t->en[x];
    const int dg = t->dep[g];
    const int dc = left ? U : -(U + X), du = left ? -C : C + X, dx = left ? 0 : C - U;
    for (int v = 0; v < t->next; v++) {                       /* one lane per node */
        if (!m_attached(t, v)) { continue; }
        const int in_c = t->st[v] >= ca && t->en[v] <= cb;
        const int in_u = t->st[v] >= ua && t->en[v] <= ub;
        const int in_x = t->st[v] >= xa && t->en[v] <= xb;
        if (v == p || v == g || !(in_c | in_u | in_x)) { continue; }
        if (in_c) {
            const int ls = t->dep[v] - dg - 2;                /* bits below c */
            if (v < t->leaves) {
                const uint64_t G = t->code[v] >> (t->dep[v] - dg);
                const uint64_t S = ls > 0 ? t->code[v] & ((1ull << ls) - 1) : 0;
                t->code[v] = (((G << 1) | (left ? 1u : 0u)) << ls) | S;
            }
            t->dep[v] -= 1; t->st[v] += dc; t->en[v] += dc;
        } else if (in_u) {
            const int ls = t->dep[v] - dg - 1;                /* bits below u */
            if (v < t->leaves) {
                const uint64_t G = t->code[v] >> (t->dep[v] - dg);
                const uint64_t S = ls > 0 ? t->code[v] & ((1ull << ls) - 1) : 0;
                t->code[v] = (((G << 2) | (left ? 1u : 3u)) << ls) | S;
            }
            t->dep[v] += 1; t->st[v] += du; t->en[v] += du;
        } else {
            t->st[v] += dx; t->en[v] += dx;
        }
    }
    /* p now holds x and u */
    t->st[p] = left ? xa : xa + dx;
    t->en[p] = t->st[p] + X + U;
    m_partners(t, t->st[g], t->en[g]);
}

/* ---- the reference sequence (huffman.h:64-147), links by one lane, everything else flat --- */
static void m_sum(mtree* t, int i) {
    const int l = t->lo[i], r = t->hi[i];
    t->freq[i] = (l != M_NIL ? t->freq[l] : 0) + (r != M_NIL ? t->freq[r] : 0);
}

static int m_order_pair(mtree* t, int i) {                    /* huffman.h:64-86 */
    const int p = t->up[i];
    if (p == M_NIL) { return i; }
    const int l = t->lo[p], r = t->hi[p];
    if (l != M_NIL && r != M_NIL && t->freq[l] > t->freq[r]) {
        t->st_swaps++;
        t->lo[p] = r; t->hi[p] = l;
        m_swap_fix(t, p);
        m_relabel_mark(t, p);
        return i == l ? r : l;
    }
    return i;
}

static void m_changed(mtree* t, int start) {                  /* huffman.h:130-147 + :98-128 */
    int pend_p[2 * M_MAXN], pend_c[2 * M_MAXN], sp = 0;
    int i = start;
    for (;;) {
        const int p = t->up[i];
        if (p == M_NIL) { m_sum(t, i); break; }
        m_sum(t, p);
        i = m_order_pair(t, i);
        pend_p[sp] = p; pend_c[sp] = i; sp++;
        i = p;
    }
    while (sp > 0) {
        sp--;
        const int p = pend_p[sp], c = pend_c[sp];
        if (t->up[p] == M_NIL || c != t->hi[p]) { continue; }
        const int par = t->up[c], g = t->up[par];
        const int left = par == t->lo[g];
        const int uncle = left ? t->hi[g] : t->lo[g];
        if (!(t->freq[c] > t->freq[uncle])) { continue; }
        t->st_moves++;
        t->up[c] = g;
        if (left) { t->hi[g] = c; } else { t->lo[g] = c; }
        t->hi[par] = uncle;
        t->up[uncle] = par;
        m_promote_fix(t, g, par, c, uncle, left);
        m_sum(t, par);
        m_sum(t, g);
        (void)m_order_pair(t, c);
        (void)m_order_pair(t, uncle);
        (void)m_order_pair(t, par);
        m_relabel_mark(t, g);
        i = g;
        for (;;) {
            const int q = t->up[i];
            if (q == M_NIL) { m_sum(t, i); break; }
            m_sum(t, q);
            i = m_order_pair(t, i);
            pend_p[sp] = q; pend_c[sp] = i; sp++;
            i = q;
        }
    }
}

/* everything from the links, by a walk from the root: used while the tree is tiny (the leaf
 * hangs straight under the root, huffman.h:156-173) */
static void m_rebuild(mtree* t) {
    int stack[M_MAXN], sp = 0, pos = 0;
    const int root = t->leaves;
    t->dep[root] = 0;
    /* iterative DFS, lo first; intervals closed on the way back */
    int order[2 * M_MAXN], n_order = 0;
    stack[sp++] = root;
    while (sp > 0) {
        const int v = stack[--sp];
        if (v < 0) { const int w = -v - 1; t->en[w] = pos; continue; }
        order[n_order++] = v;
        t->st[v] = pos;
        if (v < t->leaves) { pos++; t->en[v] = pos; continue; }
        stack[sp++] = -(v + 1);
        if (t->hi[v] != M_NIL) { stack[sp++] = t->hi[v]; }
        if (t->lo[v] != M_NIL) { stack[sp++] = t->lo[v]; }
    }
    /* depths and codes top-down in visiting order */
    uint64_t icode[M_MAXN];
    icode[root] = 0;
    for (int k = 0; k < n_order; k++) {
        const int v = order[k];
        if (v == root) { continue; }
        const int p = t->up[v];
        t->dep[v] = t->dep[p] + 1;
        icode[v] = (icode[p] << 1) | (t->hi[p] == v ? 1u : 0u);
        if (v < t->leaves) { t->code[v] = icode[v]; }
    }
    m_partners(t, 0, M_MAXN);
}

static int m_insert(mtree* t, int i) {                        /* huffman.h:149-216 */
    int ok = 1, at = t->leaves;
    t->freq[i] = 1;
    while (at >= t->leaves) {
        if (t->hi[at] == M_NIL) { t->hi[at] = i; t->up[i] = at; break; }
        if (t->lo[at] == M_NIL) { t->lo[at] = i; t->up[i] = at; break; }
        at = t->lo[at];
    }
    if (at >= t->leaves) {                                    /* under an internal node: the root, early on */
        t->freq[at]++;
        m_rebuild(t);
        i = m_order_pair(t, i);
    } else if (t->next >= t->idend) {
        ok = 0;
        t->complete = 1;
    } else {
        const int fresh = t->next++, above = t->up[at];
        const int q = t->st[at];
        /* every position behind q moves one to the right (flat pass), then the three nodes */
        for (int v = 0; v < fresh; v++) {
            if (!m_attached(t, v)) { continue; }
            if (t->st[v] > q) { t->st[v]++; }
            if (t->en[v] > q) { t->en[v]++; }
        }
        t->freq[fresh] = t->freq[at];
        t->lo[fresh] = at; t->hi[fresh] = i; t->up[fresh] = above;
        t->dep[fresh] = t->dep[at];
        if (above != M_NIL) {
            if (t->lo[above] == at) { t->lo[above] = fresh; } else { t->hi[above] = fresh; }
        }
        t->up[at] = fresh; t->up[i] = fresh;
        t->st[fresh] = q; t->en[fresh] = q + 2;
        t->st[at] = q; t->en[at] = q + 1;
        t->st[i] = q + 1; t->en[i] = q + 2;
        t->dep[at] = t->dep[fresh] + 1; t->dep[i] = t->dep[fresh] + 1;
        t->code[i] = (t->code[at] << 1) | 1u;
        t->code[at] = t->code[at] << 1;
        m_sum(t, fresh);
        /* tests: fresh takes over at's; at and i are a new pair; fresh's sibling, if it is a hi
         * child's parent's ..., keeps its own (its uncle did not change): only nodes whose
         * sibling or uncle was `at` point at a stale id -> recompute inside the parent of fresh */
        m_partners(t, above != M_NIL ? t->st[above] : 0, above != M_NIL ? t->en[above] : M_MAXN);
        at = fresh;
    }
    m_changed(t, i);
    m_relabel_mark(t, at);
    return ok;
}

static void m_bump(mtree* t, int i) {                         /* huffman.h:218-235 */
    if (t->up[i] == M_NIL) { (void)m_insert(t, i); }
    else if (!t->complete && t->mark < 63) { t->freq[i]++; m_changed(t, i); }
    else { t->complete = 1; }
}

/* ---- the batch ------------------------------------------------------------------------------ */
/* symbols sym[0..m): how many leading ones may be applied together; codes/depths filled for all */
static int m_batch(mtree* t, const int* sym, int m, uint64_t* codes, int* deps) {
    static uint32_t hist[M_MAXN + 1], P[M_MAXN + 2];
    int n_pos = t->en[t->leaves];
    for (int k = 0; k <= n_pos; k++) { hist[k] = 0; }
    for (int j = 0; j < m; j++) {                             /* lane = token */
        codes[j] = t->code[sym[j]]; deps[j] = t->dep[sym[j]];
        hist[t->st[sym[j]]]++;
    }
    P[0] = 0;
    for (int k = 0; k < n_pos; k++) { P[k + 1] = P[k] + hist[k]; }
    int ok = m;
    for (int v = 0; v < t->next; v++) {                       /* lane = node */
        if (!m_attached(t, v) || v == t->leaves) { continue; }
        const uint32_t n = P[t->en[v]] - P[t->st[v]];
        if (n == 0 || t->partner[v] == M_NIL) { continue; }
        const uint32_t bound = t->freq[t->partner[v]], f = t->freq[v];
        if (f + n <= bound) { continue; }
        const uint32_t allowed = bound > f ? bound - f : 0;  /* tokens through v that may pass */
        uint32_t seen = 0;
        for (int j = 0; j < m; j++) {                         /* the (allowed+1)-th one through v */
            const int q = t->st[sym[j]];
            if (q >= t->st[v] && q < t->en[v]) {
                if (seen == allowed) { if (j < ok) { ok = j; } break; }
                seen++;
            }
        }
    }
    /* apply the prefix: counts of every node it touches */
    for (int k = 0; k <= n_pos; k++) { hist[k] = 0; }
    for (int j = 0; j < ok; j++) { hist[t->st[sym[j]]]++; }
    for (int k = 0; k < n_pos; k++) { P[k + 1] = P[k] + hist[k]; }
    for (int v = 0; v < t->next; v++) {
        if (!m_attached(t, v)) { continue; }
        t->freq[v] += P[t->en[v]] - P[t->st[v]];
    }
    return ok;
}

/* ---- lockstep comparison with the oracle's tree ----------------------------------------------- */
static uint64_t stream_code(uint64_t path, int bits) {        /* LSB-first path -> first branch on top */
    uint64_t c = 0;
    for (int k = 0; k < bits; k++) { c = (c << 1) | ((path >> k) & 1u); }
    return c;
}

static int fail(const char* what, int a, long long x, long long y) {
    fprintf(stderr, "MISMATCH %s at %d: model %lld oracle %lld\n", what, a, x, y);
    return 1;
}

/* same shape, counts, depths, codes, mark; intervals and partners consistent with the shape */
static int compare(const mtree* t, const tree* o) {
    int sm[M_MAXN], so[M_MAXN], sp = 0, pos = 0, bad = 0;
    const int oroot = 2 * o->n - 2;
    if (t->mark != o->depth) { return fail("mark", 0, t->mark, o->depth); }
    if (t->complete != o->complete) { return fail("complete", 0, t->complete, o->complete); }
    if (t->st_updates != o->st_updates) { return fail("updates", 0, (long long)t->st_updates, (long long)o->st_updates); }
    if (t->st_swaps != o->st_swaps) { return fail("swaps", 0, (long long)t->st_swaps, (long long)o->st_swaps); }
    if (t->st_moves != o->st_moves) { return fail("moves", 0, (long long)t->st_moves, (long long)o->st_moves); }
    sm[0] = t->leaves; so[0] = oroot; sp = 1;
    while (sp > 0 && !bad) {
        sp--;
        const int v = sm[sp], w = so[sp];
        if (v != t->leaves && (uint64_t)t->freq[v] != o->freq[w]) { bad = fail("freq", v, t->freq[v], (long long)o->freq[w]); }
        if (t->dep[v] != o->bits[w]) { bad = fail("depth", v, t->dep[v], o->bits[w]); }
        if (t->st[v] != pos) { bad = fail("interval start", v, t->st[v], pos); }
        if (v < t->leaves) {
            if (v != w) { bad = fail("leaf id", v, v, w); }
            if (t->code[v] != stream_code(o->path[w], o->bits[w])) { bad = fail("code", v, (long long)t->code[v], (long long)stream_code(o->path[w], o->bits[w])); }
            if (t->en[v] != pos + 1) { bad = fail("leaf interval", v, t->en[v], pos + 1); }
            pos++;
            continue;
        }
        /* children: hi pushed first so that lo is visited first */
        const int ml = t->lo[v], mh = t->hi[v], ol = o->lo[w], oh = o->hi[w];
        if ((ml == M_NIL) != (ol < 0) || (mh == M_NIL) != (oh < 0)) { bad = fail("child presence", v, ml, ol); break; }
        if (mh != M_NIL) { if (t->up[mh] != v) { bad = fail("up", mh, t->up[mh], v); } sm[sp] = mh; so[sp] = oh; sp++; }
        if (ml != M_NIL) { if (t->up[ml] != v) { bad = fail("up", ml, t->up[ml], v); } sm[sp] = ml; so[sp] = ol; sp++; }
    }
    if (bad) { return 1; }
    /* interval ends + partners, from the links */
    for (int v = 0; v < t->next; v++) {
        if (!m_attached(t, v) || v < t->leaves) { continue; }
        const int l = t->lo[v], h = t->hi[v];
        const int a = l != M_NIL ? t->st[l] : t->st[h], b = h != M_NIL ? t->en[h] : t->en[l];
        if (t->st[v] != a || t->en[v] != b) { return fail("interval", v, t->en[v], b); }
        if (l != M_NIL && h != M_NIL && t->en[l] != t->st[h]) { return fail("children adjacent", v, t->en[l], t->st[h]); }
    }
    for (int v = 0; v < t->next; v++) {
        if (!m_attached(t, v) || v == t->leaves) { continue; }
        const int p = t->up[v];
        int want = M_NIL;
        if (t->hi[p] != v) { want = t->hi[p]; }
        else if (t->up[p] != M_NIL) { const int g = t->up[p]; want = t->lo[g] == p ? t->hi[g] : t->lo[g]; }
        if (t->partner[v] != want) { return fail("partner", v, t->partner[v], want); }
    }
    return 0;
}

/* drive both with a symbol sequence: batches of up to `batch` symbols through m_batch, the
 * symbol a batch stops at (and unseen symbols) through the exact path.  Returns 0 when the
 * model and the oracle agree after every step and on every code. */
int range_model_run(int32_t n_ref, const int32_t* symbols, uint64_t count, int batch,
                    uint64_t* out_stats /* batches, batched symbols, exact symbols, compares */) {
    tree* o = (tree*)malloc(sizeof(tree));
    mtree* t = (mtree*)malloc(sizeof(mtree));
    if (o == NULL || t == NULL) { return ENOMEM; }
    tree_init(o, n_ref);
    if (n_ref == 512) { m_init(t, 288, 576, 512); } else { m_init(t, n_ref, 2 * n_ref, n_ref); }
    uint64_t k = 0, n_batches = 0, n_batched = 0, n_exact = 0, n_cmp = 0;
    int rc = 0;
    while (k < count && rc == 0) {
        /* the batch: seen symbols only, while the tree takes updates */
        int sym[M_BATCH], m = 0;
        while (m < batch && k + (uint64_t)m < count && t->up[symbols[k + m]] != M_NIL) { sym[m] = symbols[k + m]; m++; }
        int done = 0;
        if (m > 0 && !t->complete && t->mark < 63) {
            uint64_t codes[M_BATCH]; int deps[M_BATCH];
            done = m_batch(t, sym, m, codes, deps);
            for (int j = 0; j < done && rc == 0; j++) {      /* the oracle, one at a time */
                const int s = sym[j];
                if (codes[j] != stream_code(o->path[s], o->bits[s]) || deps[j] != o->bits[s]) {
                    rc = fail("batched code", s, (long long)codes[j], (long long)stream_code(o->path[s], o->bits[s]));
                }
                tree_bump(o, s);
            }
            if (done > 0) { n_batches++; n_batched += (uint64_t)done; }
            k += (uint64_t)done;
            if (rc == 0 && done > 0) { rc = compare(t, o); n_cmp++; }
        }
        if (rc == 0 && k < count && (done < m || m == 0 || t->complete || t->mark >= 63)) {
            const int s = symbols[k];
            if (t->up[s] != M_NIL && (t->code[s] != stream_code(o->path[s], o->bits[s]))) {
                rc = fail("exact code", s, (long long)t->code[s], (long long)stream_code(o->path[s], o->bits[s]));
            }
            m_bump(t, s);
            tree_bump(o, s);
            k++; n_exact++;
            if (rc == 0) { rc = compare(t, o); n_cmp++; }
        }
    }
    if (out_stats != NULL) { out_stats[0] = n_batches; out_stats[1] = n_batched; out_stats[2] = n_exact; out_stats[3] = n_cmp; }
    free(o); free(t);
    return rc;
}
