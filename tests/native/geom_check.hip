/*
 * Host-side check of the traceback-store geometry the kernels use (badread_amd/csrc/brx_align.h: brx_make_geom,
 * brx_stored, brx_jrep, brx_tb_units are __host__ __device__): compiled with hipcc, run on the CPU by
 * tests/test_geom_host.py.  For every case it walks all (superblock, column) cells of the band and verifies
 *   1. no two stored cells share a (store row, slot) address, and every address lies inside brx_tb_units();
 *   2. every superblock that intersects rows [c(j) - H, c(j) + H] of column j is stored (the window guarantee);
 *   3. the windowed store is never larger than the full one, and the full store keeps every band cell.
 */
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

#include "../../badread_amd/csrc/brx_align.h"
#include "../../badread_amd/csrc/brx_quad.h"

static int host_jfirst(const BrxGeom &g, int s) { int j = g.R * s - g.dhi + 1; return j < 1 ? 1 : j; }
static int host_jlast(const BrxGeom &g, int s) { long long j = (long long)g.R * (s + 1) - g.dlo; return j > g.T ? g.T : (int)j; }

static int check(int Q, int T, int k, int hmul) {
    const BrxGeom g = brx_make_geom(Q, T, k, hmul);
    const BrxGeom full = brx_make_geom(Q, T, k, 0);
    if (g.G == 0) return full.G == 0 ? 0 : 1;
    if (brx_tb_units(g) > brx_tb_units(full)) { printf("windowed store larger than full: Q=%d T=%d k=%d\n", Q, T, k); return 1; }
    const uint64_t units = brx_tb_units(g);
    std::unordered_set<uint64_t> seen;
    for (int s = 0; s < g.NS; ++s) {
        for (int j = host_jfirst(g, s); j <= host_jlast(g, s); ++j) {
            const bool st = brx_stored(g, s, brx_jrep(g, s, j));
            /* the forward passes decide per loop trip: tau = (j + s - 1) / U is the trip that computes column j of superblock s
               (time j + s), and their keep test must be the traceback's brx_stored for every column of the trip (corner trips too) */
            if (g.U > 1) {
                const int tau = (j + s - 1) / g.U;
                if (j < g.U * tau + 1 - s || j > g.U * tau + g.U - s) { printf("trip of column: Q=%d T=%d k=%d s=%d j=%d\n", Q, T, k, s, j); return 1; }
                const bool fw = brx_keep_trip(g.slope, g.R * s + g.H + g.R - 1, (uint32_t)(2 * g.H + g.R - 1), brx_jrep_trip(g.U, tau, s));
                if (fw != st) { printf("forward keep != traceback stored: Q=%d T=%d k=%d s=%d j=%d\n", Q, T, k, s, j); return 1; }
            }
            if (g.H == BRX_H_ALL && !st) { printf("full store drops a band cell: Q=%d T=%d k=%d s=%d j=%d\n", Q, T, k, s, j); return 1; }
            if (g.H != BRX_H_ALL) {
                const long long c = (long long)(((uint64_t)(uint32_t)brx_jrep(g, s, j) * (uint64_t)g.slope) >> 20);
                const long long lo = (long long)g.R * s, hi = lo + g.R - 1;
                const bool meets = hi >= c - g.H && lo <= c + g.H;
                if (meets != st) { printf("window predicate: Q=%d T=%d k=%d s=%d j=%d meets=%d stored=%d\n", Q, T, k, s, j, meets, st); return 1; }
            }
            if (!st) continue;
            const uint64_t addr = ((uint64_t)(j + g.K * s) * (uint64_t)g.WSp + (uint64_t)(s % g.WSp)) * (uint64_t)g.G;
            if (addr + (uint64_t)g.G > units) { printf("address outside the store: Q=%d T=%d k=%d s=%d j=%d\n", Q, T, k, s, j); return 1; }
            if (!seen.insert(addr).second) { printf("two cells share a slot: Q=%d T=%d k=%d hmul=%d s=%d j=%d WSp=%d\n", Q, T, k, hmul, s, j, g.WSp); return 1; }
        }
    }
    return 0;
}

/* four alignments side by side in one slab (brx_quad.h): rows of 4 x WSq slots; no two cells of the group share an address and
   every address lies inside brx_quad_units() */
static int quads_checked = 0;
static int check_quad(const int (*qtk)[3], int n, int hmul) {
    BrxGeom g4[4];
    int wsq = 1, G = 0;
    for (int i = 0; i < n; ++i) {
        g4[i] = brx_make_geom_quad(qtk[i][0], qtk[i][1], qtk[i][2], hmul);
        if (g4[i].G == 0) return 0;                       /* not a quad geometry: nothing to check */
        if (G && g4[i].G != G) return 0;                  /* lists are class-pure */
        G = g4[i].G;
        if (g4[i].WSp > wsq) wsq = g4[i].WSp;
    }
    ++quads_checked;
    const uint64_t units = brx_quad_units(g4, n);
    uint64_t peq = 0;
    for (int i = 0; i < n; ++i) peq += brx_peq_units(g4[i]);
    std::unordered_set<uint64_t> seen;
    for (int i = 0; i < n; ++i) {
        BrxGeom g = g4[i];
        if ((long long)g.dhi - g.dlo + 1 > (long long)BRX_QUAD_SPAN * g.R) { printf("quad band too wide\n"); return 1; }
        g.WSp = wsq; g.WSrow = 4 * wsq; g.slot0 = i * wsq;
        for (int s = 0; s < g.NS; ++s) {
            /* a lane's next superblock (s + 16) must start after this one has ended: whole trips */
            if (s + BRX_QUAD_LW < g.NS && host_jlast(g, s) >= host_jfirst(g, s) && host_jlast(g, s + BRX_QUAD_LW) >= host_jfirst(g, s + BRX_QUAD_LW)) {
                const int tl = (host_jlast(g, s) + s - 1) / g.U, tf = (host_jfirst(g, s + BRX_QUAD_LW) + s + BRX_QUAD_LW - 1) / g.U;
                if (tf <= tl) { printf("row hop overlaps: Q=%d T=%d k=%d s=%d\n", g.Q, g.T, qtk[i][2], s); return 1; }
            }
            for (int j = host_jfirst(g, s); j <= host_jlast(g, s); ++j) {
                if (!brx_stored(g, s, brx_jrep(g, s, j))) continue;
                const uint64_t addr = ((uint64_t)(j + g.K * s) * (uint64_t)g.WSrow + (uint64_t)(g.slot0 + s % g.WSp)) * (uint64_t)g.G;
                if (addr + (uint64_t)g.G > units - peq) { printf("quad address outside the rows: Q=%d T=%d s=%d j=%d\n", g.Q, g.T, s, j); return 1; }
                if (!seen.insert(addr).second) { printf("quad cells share a slot: Q=%d T=%d s=%d j=%d\n", g.Q, g.T, s, j); return 1; }
            }
        }
    }
    return 0;
}

int main() {
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](uint32_t n) { state = state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)((state >> 33) % n); };
    const int lens[] = {1, 2, 31, 32, 33, 64, 100, 500, 1000, 1014, 3000, 8000, 20000, 45000};
    const int hm[] = {0, 1, 2, 4, -1};
    int cases = 0;
    for (int it = 0; it < 1500; ++it) {
        const int T = lens[rnd(sizeof(lens) / sizeof(lens[0]))];
        int Q = (int)((double)T * (0.85 + 0.3 * (rnd(1000) / 1000.0))) + (int)rnd(5) - 2;
        if (Q < 1) Q = 1;
        const int ad = Q > T ? Q - T : T - Q;
        const int rates[] = {0, 1, 5, 12, 30};
        int k = ad + (int)((long long)T * rates[rnd(5)] / 100);
        if (T > 20000 && k > 4000) k = 4000 + (int)rnd(3000);          /* keep the exhaustive walk short */
        if (check(Q, T, k, hm[rnd(5)])) return 1;
        ++cases;
    }
    /* corner cases */
    const int fixed[][4] = {{1, 1, 0, 2}, {7, 6, 7, 2}, {1000, 1794, 1024, 2}, {1000, 1794, 2048, 2}, {15000, 15011, 1940, 2},
                            {60000, 60012, 2669, 2}, {128, 28, 128, 2}, {32, 4000, 3968, 2}, {4000, 32, 3968, 2}};
    for (const auto &f : fixed) { if (check(f[0], f[1], f[2], f[3])) return 1; ++cases; }
    /* groups of four for k_fin_quad: similar lengths, bands of 1 .. 13 superblocks of one or two words */
    for (int it = 0; it < 300; ++it) {
        int qtk[4][3];
        const int n = 1 + (int)rnd(4);
        const int base = lens[4 + rnd(8)];
        const int two = (int)rnd(2);
        for (int i = 0; i < n; ++i) {
            const int T = base - (int)rnd(base / 4 + 1);
            int Q = T + (int)rnd(41) - 20;
            if (Q < 1) Q = 1;
            const int ad = Q > T ? Q - T : T - Q;
            int k = two ? 418 + (int)rnd(400) : ad + (int)rnd(395 - ad);
            qtk[i][0] = Q; qtk[i][1] = T; qtk[i][2] = k;
        }
        if (check_quad(qtk, n, hm[rnd(5)])) return 1;
        ++cases;
    }
    if (quads_checked < 200) { printf("only %d groups of four checked\n", quads_checked); return 1; }
    printf("ok %d cases, %d groups of four\n", cases, quads_checked);
    return 0;
}
