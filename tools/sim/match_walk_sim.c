// Walk simulator for the match stage (development tool, CPU only; not part of the library or the oracle).
//
// Replays, on a real stream window, what k_match5 does per 64-slot group of the hash-sorted order -- the filter masks of
// every lane (V, A4..A7 from the same 6/5-bit keys), the newest-first walk with its refinement -- and counts scorings and
// rounds (a round = one trip of the wave through the candidate loop: as many as the busiest lane needs), so that other
// schedules of the same work can be costed before any of them is built:
//   current      rounds of the newest 32 + rounds of the other 96, per group
//   pooled       the other 96 (optionally: whatever is left after R rounds in place) taken over by work items that are
//                refilled as lanes finish; items live G groups at most (the ring)
//   more filters extra key levels (bytes 7, 8, ...)
// and the demand side: which positions the exact parse visits (tables from the same walk).
//
//   gcc -O2 -o /tmp/match_walk_sim tools/sim/match_walk_sim.c && /tmp/match_walk_sim stream.bin [window_offset] [window_len]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

enum { MAX_DIST = 32506, MAX_MATCH = 258, CHAIN = 128, QCHAIN = 32, NICE = 128, GOOD = 8, LAZY = 16, TOO_FAR = 4096 };

static u32 hash_of(const u8 *b) { return ((b[0] << 10) ^ (b[1] << 5) ^ b[2]) & 0x7fff; }
static u32 umul24(u32 a, u32 b) { return (a & 0xffffff) * (b & 0xffffff); }
static u32 h24(u32 x) { return (umul24(x, 0x9E3779u) >> 19) & 31; }
#ifndef KEYS
#define KEYS 1            // 0: the keys of rounds 2-4, 1: round 5 (one multiplication for the first three), 9: a strong mixer (what 6 + 5 + 5 + 5 bits can do at best)
#endif
static u32 mix32(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
static void keys(const u8 *b, u32 k[6])
{
    const u32 e1 = b[3] | (b[4] << 8) | (b[5] << 16) | ((u32)b[6] << 24);
#if KEYS == 0
    k[0] = (umul24(e1 & 0xff, 0x9E3779u) >> 18) & 63;
    k[1] = h24(e1 & 0xffff);
    k[2] = h24(e1 & 0xffffff);
    k[3] = h24((e1 ^ (e1 >> 11)) & 0xffffff);
#elif KEYS == 1
    { const u32 p = umul24(e1, 0x9E3779u); k[0] = (p >> 2) & 63; k[1] = (p >> 11) & 31; k[2] = (p >> 19) & 31; }
    k[3] = h24((e1 ^ (e1 >> 11)) & 0xffffff);
#elif KEYS == 2
    { const u32 p = umul24(e1, 0x9E3779u); k[0] = (p >> 2) & 63; k[1] = (p >> 11) & 31; k[2] = (p >> 19) & 31; }
    { const u32 q = umul24(e1 ^ (e1 >> 12), 0x9E3779u); k[3] = (q >> 19) & 31; }
#elif KEYS == 3
    { const u32 p = umul24(e1, 0x9E3779u); k[0] = (p >> 2) & 63; k[1] = (p >> 11) & 31; k[2] = (p >> 19) & 31; }
    { const u32 q = (e1 >> 8) * 0x9E3779B1u; k[3] = ((q >> 27) ^ k[0]) & 31; }
#elif KEYS == 9
    k[0] = mix32(e1 & 0xff) & 63; k[1] = mix32(e1 & 0xffff) & 31; k[2] = mix32(e1 & 0xffffff) & 31; k[3] = mix32(e1) & 31;
#endif
    k[4] = b[7] & 31;                                    // extra levels (what-if): byte 7 (5 bits, exact within them), byte 8
    k[5] = (b[8] ^ (b[8] >> 3)) & 31;
}
static u32 lcp(const u8 *a, const u8 *b, u32 maxlen)
{
    u32 l = 0;
    while (l < maxlen && a[l] == b[l]) l++;
    return l;
}

// the masks are ANDed level after level: a candidate passes level `need` when every EXISTING table up to it has its key equal
static int levok(u8 bits, u32 need, int levmask) { for (u32 l = 0; l < need; l++) if (((levmask >> l) & 1) && !((bits >> l) & 1)) return 0; return 1; }

typedef struct { u32 n1, n2; u32 best, bdist, qbest, qdist; } LaneRes;

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s stream.bin [offset] [len] [levels=4]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    long off = argc > 2 ? atol(argv[2]) : 0, W = argc > 3 ? atol(argv[3]) : (1 << 18);
    const int NLEV = argc > 5 ? atoi(argv[5]) : 4;       // key levels considered (5, 6: what-if tables for bytes 7 and 8)
    const int LEVMASK = argc > 4 ? atoi(argv[4]) : 15;    // which of the four key tables exist: bit d = the table of prefix length 4 + d (15 = as built)
    if (off + W > fsz) W = fsz - off;
    u8 *s = malloc(W + 512);
    memset(s, 0, W + 512);
    fseek(f, off, SEEK_SET);
    if (fread(s, 1, W, f) != (size_t)W) { perror("read"); return 2; }
    fclose(f);
    const u32 n = (u32)W, wlen = n - 2;
    // sorted order: stable by (hash, position)
    u32 *cnt = calloc(32769, 4), *sorted = malloc(4 * wlen), *slot_of = malloc(4 * wlen);
    for (u32 p = 0; p < wlen; p++) cnt[hash_of(s + p) + 1]++;
    for (int h = 0; h < 32768; h++) cnt[h + 1] += cnt[h];
    for (u32 p = 0; p < wlen; p++) sorted[cnt[hash_of(s + p)]++] = p;
    for (u32 i = 0; i < wlen; i++) slot_of[sorted[i]] = i;
    u32 *tf = calloc(n + 4, 4), *tq = calloc(n + 4, 4);          // (len << 16 | dist)
    const u32 ngroups = (wlen + 63) / 64;
    u64 tot1 = 0, tot2 = 0, rounds1 = 0, rounds2 = 0, lanes = 0, hist1[64] = {0}, hist2[64] = {0}, wasted = 0;
    u64 act1[40] = {0}, act2[40] = {0}, merged[4] = {0};
    u64 hy_bad = 0, hy_tot = 0, hy_rounds = 0, hy_act[40] = {0}; u32 hy_n[64];
    u64 oor1 = 0, oor2 = 0, td_fp_r1 = 0, td_oor_r1 = 0, td_l7_r1 = 0, td_fp_groups = 0; int g_fp = 0;
    u64 td_bad = 0, td_badq = 0, td_tot1 = 0, td_tot2 = 0, td_rounds1 = 0, td_rounds2 = 0, td_act1[40] = {0}, td_act2[40] = {0};
    u32 td_n1[64], td_n2[64];
    // pooled what-ifs: in-place rounds R for the newest word, everything else as items
    enum { NR = 4 };
    u64 inplace_rounds[NR] = {0}, items[NR] = {0}, item_scorings[NR] = {0}, item_max[NR] = {0};
    LaneRes *res = malloc(sizeof(LaneRes) * 64);
    u32 (*seq)[160] = malloc(64 * 160 * 4);                 // per lane: scorings in walk order (candidate index j), for the what-ifs
    for (u32 g = 0; g < ngroups; g++) {
        u32 gmax1 = 0, gmax2 = 0; g_fp = 0;
        u32 nseq[64];
        for (int lane = 0; lane < 64; lane++) {
            const u32 i = g * 64 + lane;
            nseq[lane] = 0; td_n1[lane] = td_n2[lane] = 0; hy_n[lane] = 0;
            res[lane].n1 = res[lane].n2 = 0;
            if (i >= wlen) continue;
            const u32 p = sorted[i];
            const u8 *me = s + p;
            const u32 h = hash_of(me);
            u32 nbv = 0;
            while (nbv < CHAIN && i > nbv && hash_of(s + sorted[i - nbv - 1]) == h) nbv++;
            const u32 look = n - p, maxlen = look < MAX_MATCH ? look : MAX_MATCH, nice = NICE < look ? NICE : look;
            u32 mk[6];
            keys(me, mk);
            // level of every candidate: how many key levels agree in a row (0..NLEV)
            u8 lev_bits[CHAIN + 1];                              // bit d: key d equal
            for (u32 j = 1; j <= nbv; j++) {
                u32 ck[6];
                keys(s + sorted[i - j], ck);
                u8 b = 0;
                for (int l = 0; l < NLEV; l++) if (ck[l] == mk[l]) b |= 1 << l;
                lev_bits[j] = b;
            }
            u8 levj_h[CHAIN + 1];
            for (u32 j = 1; j <= nbv; j++) { u32 l = 0; while (l < (u32)NLEV && ((lev_bits[j] >> l) & 1)) l++; levj_h[j] = (u8)(3 + l); }
            u32 best = 2, bdist = 0, qbest = 2, qdist = 0;
            int stop = 0, qset = 0;
            lanes++;
            for (u32 j = 1; j <= nbv && !stop; j++) {
                if (j == QCHAIN + 1) { qbest = best; qdist = bdist; qset = 1; }
                // is candidate j let through at the current best?  level wanted: best>=3 -> A4 (lev>=1), >=4 -> A5, ... capped at NLEV;
                // a table that does not exist is replaced by the deepest existing one below it
                u32 need = best >= 3 ? best - 2 : 0;
                if (need > (u32)NLEV) need = NLEV;
                while (need && !((LEVMASK >> (need - 1)) & 1)) need--;
                if (need && !levok(lev_bits[j], need, LEVMASK)) continue;
                // scored
                if (j <= QCHAIN) res[lane].n1++; else res[lane].n2++;
                seq[lane][nseq[lane]++] = j;
                const u32 c = sorted[i - j];
                // zlib: the head of the chain may be MAX_DIST away, the others must be nearer; position 0 is NIL
                const u32 lim = j == 1 ? (p > MAX_DIST + 1 ? p - MAX_DIST - 1 : 0) : (p > MAX_DIST ? p - MAX_DIST : 0);
                if (!(c > lim)) { stop = 1; if (j > QCHAIN) oor2++; else oor1++; break; }
                if (memcmp(s + c, me, 3)) { wasted++; continue; }                       // same hash, other bytes
                const u32 len = lcp(s + c, me, maxlen);
                if (len > best) { best = len; bdist = p - c; if (len >= nice) stop = 1; }
                else wasted++;
            }
            if (!qset) { qbest = best; qdist = bdist; }             // the walk ended inside the newest QCHAIN candidates
            res[lane].best = best; res[lane].bdist = bdist; res[lane].qbest = qbest; res[lane].qdist = qdist;
            {   // what-if: TOP-DOWN walks.  Word by word (newest first), inside a word the candidates of the highest non-empty filter level
                // first: one that verifies at its level d < 7 ends the word at once (nothing in the word can be longer, ties go to
                // the newer one), level 7 is walked newest-first as before, a false positive leaves the candidates that can still tie
                // or win.  Same results (checked below), fewer scorings per lane -- and per group?
                u32 tb_len = 2, tb_pos = 0, tq_len = 2, tq_pos = 0, td1 = 0, td2 = 0;
                int dead = 0, tstop = 0;
                u8 levj[CHAIN + 1];
                for (u32 j = 1; j <= nbv; j++) { u32 l = 0; while (l < (u32)NLEV && ((lev_bits[j] >> l) & 1)) l++; levj[j] = (u8)(3 + l); }
                for (int w = 3; w >= 0 && !dead && !tstop; w--) {
                    const u32 j0 = (u32)(3 - w) * 32 + 1, j1 = j0 + 31 < nbv ? j0 + 31 : nbv;
                    if (w == 2) { tq_len = tb_len; tq_pos = tb_pos; }
                    if (j0 > nbv) break;
                    u8 inR[33];
                    u32 need0 = tb_len >= 3 ? tb_len + 1 : 3;
                    if (need0 > 7) need0 = 7;
                    for (u32 j = j0; j <= j1; j++) inR[j - j0] = levj[j] >= need0;
                    for (;;) {
                        u32 dstar = 0, cj = 0;
                        for (u32 j = j0; j <= j1; j++) if (inR[j - j0] && levj[j] > dstar) { dstar = levj[j]; cj = j; }
                        if (!cj) break;
                        if (w == 3) td1++; else td2++;
                        const u32 c = sorted[i - cj];
                        const u32 lim = cj == 1 ? (p > MAX_DIST + 1 ? p - MAX_DIST - 1 : 0) : (p > MAX_DIST ? p - MAX_DIST : 0);
                        if (!(c > lim)) { for (u32 j = cj; j <= j1; j++) inR[j - j0] = 0; dead = 1; if (w == 3 && td1 == 1) { td_oor_r1++; g_fp = 1; } continue; }
                        inR[cj - j0] = 0;
                        u32 len = memcmp(s + c, me, 3) ? 0 : lcp(s + c, me, maxlen);
                        if (len >= 3 && (len > tb_len || (len == tb_len && c > tb_pos))) { tb_len = len; tb_pos = c; }
                        if (len >= nice) { tstop = 1; break; }
                        if (w == 3 && td1 == 1) { if (len < dstar) { td_fp_r1++; g_fp = 1; } else if (dstar == 7) td_l7_r1++; }
                        if (len >= dstar) {
                            if (dstar < 7) break;
                            for (u32 j = j0; j <= j1; j++) if (levj[j] < 7) inR[j - j0] = 0;
                        } else {
                            const u32 keep = tb_len >= 3 ? (tb_len > 7 ? 7 : tb_len) : 3;
                            for (u32 j = j0; j <= j1; j++) if (levj[j] < keep) inR[j - j0] = 0;
                        }
                    }
                }
                if (nbv <= 32 || 1) { if (!(nbv > 32) || 0) { } }
                if (nbv <= 32) { tq_len = tb_len; tq_pos = tb_pos; }
                else if (tstop && td2 == 0 && tq_len == 2 && tb_len > 2 && 0) { }
                // (a walk that ended inside the newest word: the quarter result is what it holds)
                if (nbv > 32 && (tstop || dead) && td2 == 0) { /* ended in word 3 or at its border */ if (tq_len == 2 && tq_pos == 0) { tq_len = tb_len; tq_pos = tb_pos; } }
                const u32 rb = best >= 3 ? best : 2, rq = qbest >= 3 ? qbest : 2;
                if (tb_len != rb || (rb >= 3 && p - tb_pos != bdist)) { td_bad++; if (getenv("TD_DEBUG")) printf("p %u nbv %u look %u: ref %u/%u q %u/%u  td %u/%u q %u/%u dead %d stop %d td1 %u td2 %u\n", p, nbv, look, best, bdist, qbest, qdist, tb_len, p - tb_pos, tq_len, p - tq_pos, dead, tstop, td1, td2); }
                if (tq_len != rq || (rq >= 3 && p - tq_pos != qdist)) td_badq++;
                td_n1[lane] = td1; td_n2[lane] = td2; td_tot1 += td1; td_tot2 += td2;
            }
            {   // what-if: HYBRID.  Round 1 of the newest word top-down (every lane scores the newest candidate of its highest non-empty
                // level); a lane whose candidate verifies below level 7 is done, one at level 7 goes on through A7; a false positive
                // (shorter than its level said) or an out-of-range candidate falls back to the newest-first walk with a floor:
                // best = len - 1 held by nobody, candidates = those that may be >= len long (the scored one among them), resp.
                // from scratch over the candidates newer than the one out of range.  Newest-first with strict improvement from there.
                u32 hy = 0, hb = 2, hpos = 0;
                const u32 j1 = nbv < 32 ? nbv : 32;
                if (j1) {
                    u32 dstar = 0, cj = 0;
                    for (u32 j = 1; j <= j1; j++) if (levj_h[j] > dstar) { dstar = levj_h[j]; cj = j; }
                    hy = 1;
                    const u32 c = sorted[i - cj];
                    const u32 lim = cj == 1 ? (p > MAX_DIST + 1 ? p - MAX_DIST - 1 : 0) : (p > MAX_DIST ? p - MAX_DIST : 0);
                    u32 floor_ = 2, jlo = 1, jhi = j1, needlev = 3; int skip_c = 0, done = 0;
                    if (!(c > lim)) { jhi = cj - 1; }
                    else {
                        const u32 len = memcmp(s + c, me, 3) ? 0 : lcp(s + c, me, maxlen);
                        if (len >= nice) { hb = len; hpos = c; done = 1; }
                        else if (len >= dstar) { hb = len; hpos = c; floor_ = len; if (dstar < 7) done = 1; else { needlev = 7; skip_c = 1; } }
                        else if (len >= 3) { floor_ = len - 1; needlev = len > 7 ? 7 : len; }
                        else { skip_c = 1; }
                    }
                    if (!done) {
                        u32 b = floor_;
                        for (u32 j = jlo; j <= jhi; j++) {
                            if (skip_c && j == cj) continue;
                            u32 need = b >= 3 ? b + 1 : 3; if (need > 7) need = 7;
                            if (b == floor_ && floor_ >= 2 && needlev > need) need = needlev;
                            if (levj_h[j] < need) continue;
                            hy++;
                            const u32 cc = sorted[i - j];
                            const u32 lm = j == 1 ? (p > MAX_DIST + 1 ? p - MAX_DIST - 1 : 0) : (p > MAX_DIST ? p - MAX_DIST : 0);
                            if (!(cc > lm)) break;
                            const u32 len = memcmp(s + cc, me, 3) ? 0 : lcp(s + cc, me, maxlen);
                            if (len > b) { b = len; hb = len; hpos = cc; if (len >= nice) break; }
                        }
                    }
                }
                const u32 rq = qbest >= 3 ? qbest : 2;
                if (look >= MAX_MATCH && (hb != rq || (rq >= 3 && p - hpos != qdist))) hy_bad++;
                hy_n[lane] = hy; hy_tot += hy;
            }
            tf[p] = best >= 3 ? (best << 16) | bdist : 0;
            tq[p] = qbest >= 3 ? (qbest << 16) | qdist : 0;
            tot1 += res[lane].n1; tot2 += res[lane].n2;
            hist1[res[lane].n1 < 63 ? res[lane].n1 : 63]++;
            hist2[res[lane].n2 < 63 ? res[lane].n2 : 63]++;
            if (res[lane].n1 > gmax1) gmax1 = res[lane].n1;
            if (res[lane].n2 > gmax2) gmax2 = res[lane].n2;
        }
        rounds1 += gmax1; rounds2 += gmax2; td_fp_groups += g_fp;
        { u32 m = 0; for (int l = 0; l < 64; l++) if (hy_n[l] > m) m = hy_n[l]; hy_rounds += m; for (u32 r = 0; r < m && r < 40; r++) for (int l = 0; l < 64; l++) if (hy_n[l] > r) hy_act[r]++; }
        { u32 m1 = 0, m2 = 0; for (int l = 0; l < 64; l++) { if (td_n1[l] > m1) m1 = td_n1[l]; if (td_n2[l] > m2) m2 = td_n2[l]; } td_rounds1 += m1; td_rounds2 += m2;
          for (u32 r = 0; r < m1 && r < 40; r++) for (int l = 0; l < 64; l++) if (td_n1[l] > r) td_act1[r]++;
          for (u32 r = 0; r < m2 && r < 40; r++) for (int l = 0; l < 64; l++) if (td_n2[l] > r) td_act2[r]++; }
        for (int R = 1; R <= 3; R++) { u32 m = 0; for (int l = 0; l < 64; l++) { const u32 left = (res[l].n1 > (u32)R ? res[l].n1 - R : 0) + res[l].n2; if (left > m) m = left; } merged[R] += (gmax1 < (u32)R ? gmax1 : (u32)R) + m; }
        for (u32 r = 0; r < gmax1 && r < 40; r++) for (int l = 0; l < 64; l++) if (res[l].n1 > r) act1[r]++;
        for (u32 r = 0; r < gmax2 && r < 40; r++) for (int l = 0; l < 64; l++) if (res[l].n2 > r) act2[r]++;
        // what-if: R rounds of the newest word in place, then every lane with anything left becomes an item
        for (int R = 0; R < NR; R++) {
            u32 rr = gmax1 < (u32)R ? gmax1 : (u32)R;
            inplace_rounds[R] += rr;
            for (int l = 0; l < 64; l++) {
                const u32 left = (res[l].n1 > (u32)R ? res[l].n1 - R : 0) + res[l].n2;
                if (left) { items[R]++; item_scorings[R] += left; if (left > item_max[R]) item_max[R] = left; }
            }
        }
    }
    printf("window %ld bytes at %ld: %u groups, %llu lanes\n", W, off, ngroups, (unsigned long long)lanes);
    printf("scorings per position: newest32 %.3f  other96 %.3f  (wasted %.3f)\n", (double)tot1 / lanes, (double)tot2 / lanes, (double)wasted / lanes);
    printf("rounds per group:      newest32 %.2f  other96 %.2f   lane use %.1f%% / %.1f%%\n", (double)rounds1 / ngroups, (double)rounds2 / ngroups,
           100.0 * tot1 / (rounds1 * 64.0), 100.0 * tot2 / (rounds2 * 64.0));
    printf("rounds per group if the newest word gets R rounds of its own and its leftovers join the loop of the other 96: R=1 %.2f  R=2 %.2f  R=3 %.2f\n", (double)merged[1] / ngroups, (double)merged[2] / ngroups, (double)merged[3] / ngroups);
    printf("TOP-DOWN: scorings per position %.3f + %.3f, rounds per group %.2f + %.2f, results differing: %llu full, %llu quarter\n", (double)td_tot1 / lanes, (double)td_tot2 / lanes, (double)td_rounds1 / ngroups, (double)td_rounds2 / ngroups, (unsigned long long)td_bad, (unsigned long long)td_badq);
    printf("TOP-DOWN active lanes per round: "); for (int r = 0; r < 8; r++) printf("%.1f ", (double)td_act1[r] / ngroups); printf(" | "); for (int r = 0; r < 8; r++) printf("%.1f ", (double)td_act2[r] / ngroups); printf("\n");
    printf("scorings that only find the candidate out of range: newest32 %.3f other96 %.3f per position\n", (double)oor1 / lanes, (double)oor2 / lanes);
    printf("TOP-DOWN round 1 of the newest word, per group: %.2f lanes with a false positive, %.2f out of range, %.2f verified at level 7; groups with a false positive or out-of-range lane: %.1f%%\n", (double)td_fp_r1 / ngroups, (double)td_oor_r1 / ngroups, (double)td_l7_r1 / ngroups, 100.0 * td_fp_groups / ngroups);
    printf("HYBRID (round 1 top-down, the rest newest-first): scorings per position %.3f, rounds per group %.2f, results differing %llu; active lanes per round: ", (double)hy_tot / lanes, (double)hy_rounds / ngroups, (unsigned long long)hy_bad); for (int r = 0; r < 8; r++) printf("%.1f ", (double)hy_act[r] / ngroups); printf("\n");
    printf("lanes by scorings (newest32): ");
    for (int k = 0; k < 12; k++) printf("%d:%.1f%% ", k, 100.0 * hist1[k] / lanes);
    printf("\nlanes by scorings (other96):  ");
    for (int k = 0; k < 12; k++) printf("%d:%.1f%% ", k, 100.0 * hist2[k] / lanes);
    printf("\nactive lanes per round (newest32): ");
    for (int r = 0; r < 10; r++) printf("%.1f ", (double)act1[r] / ngroups);
    printf("\nactive lanes per round (other96):  ");
    for (int r = 0; r < 10; r++) printf("%.1f ", (double)act2[r] / ngroups);
    printf("\n");
    for (int R = 0; R < NR; R++)
        printf("pool after %d in-place rounds: in-place %.2f rounds/group, items %.1f/group with %.2f scorings each (max %llu) = %.2f full rounds/group\n", R,
               (double)inplace_rounds[R] / ngroups, (double)items[R] / ngroups, (double)item_scorings[R] / (items[R] ? items[R] : 1),
               (unsigned long long)item_max[R], (double)item_scorings[R] / ngroups / 64.0);
    // demand side: which positions does the exact level-6 parse visit (SURVEY A.2 with the tables above)?
    {
        u64 visited = 0, vis_full = 0, vis_quarter = 0, vis_none = 0;
        u8 *vis = calloc(n + 1, 1);
        u32 p = 0, ml = 2, ms = 0, avail = 0;
        (void)ms;
        while (p < n) {
            u32 pl = ml, cand_len = 0;
            ml = 2;
            if (n - p >= 3) {
                if (pl < LAZY) {
                    const u32 e = pl >= GOOD ? tq[p] : tf[p];
                    visited++; vis[p] = 1;
                    if (pl >= GOOD) vis_quarter++; else vis_full++;
                    cand_len = e >> 16;
                    const u32 cd = e & 0xffff;
                    if (cand_len > pl) { ml = cand_len; if (ml == 3 && cd > TOO_FAR) ml = 2; }
                } else vis_none++;
            }
            if (pl >= 3 && ml <= pl) { p += pl - 1; avail = 0; ml = 2; }
            else if (avail) { p++; }
            else { avail = 1; p++; }
        }
        printf("demand: the exact parse looks up %.1f%% of the positions (full budget %.1f%%, quarter %.1f%%; %.1f%% visited without a lookup)\n",
               100.0 * visited / n, 100.0 * vis_full / n, 100.0 * vis_quarter / n, 100.0 * vis_none / n);
        // per 64-slot group of the sorted order: how many lanes are in demand
        u64 dl = 0, dgroups = 0, d_r1 = 0, d_r2 = 0;
        u64 dh[65] = {0};
        (void)d_r1; (void)d_r2;
        for (u32 g = 0; g < ngroups; g++) {
            u32 c = 0;
            for (int l = 0; l < 64; l++) { const u32 i = g * 64 + l; if (i < wlen && vis[sorted[i]]) c++; }
            dl += c; dgroups++; dh[c]++;
        }
        printf("demand per sorted group: %.1f lanes of 64 on average; groups by demanded lanes: ", (double)dl / dgroups);
        for (int c = 0; c <= 64; c += 8) { u64 a = 0; for (int k = c; k < c + 8 && k <= 64; k++) a += dh[k]; printf("%d+:%.1f%% ", c, 100.0 * a / dgroups); }
        printf("\n");
        free(vis);
    }
    // The speculative parse (k_parse_spec) starts a segment's walk AT the segment start; the true entry is where the walk of the
    // segment before ends.  What if it started W positions earlier and took its first base position at or behind the start as
    // the entry: how often is that the true one (the re-walk of the fix round then has nothing to do)?
    {
        u8 *is_base = calloc(n + 600, 1);
        #define STEP(p0, out) do { u32 p_ = (p0), len_ = tf[p_] >> 16, dist_ = tf[p_] & 0xffff; if (len_ == 3 && dist_ > TOO_FAR) len_ = 0; \
            if (len_ < 3) { (out) = p_ + 1; break; } \
            for (;;) { const u32 q_ = p_ + 1; if (q_ < n && len_ < LAZY) { const u32 d_ = len_ >= GOOD ? tq[q_] : tf[q_]; if ((d_ >> 16) > len_) { p_ = q_; len_ = d_ >> 16; continue; } } break; } \
            (out) = p_ + len_; } while (0)
        for (u32 p = 0; p < n;) { is_base[p] = 1; u32 nx; STEP(p, nx); p = nx; }
        const int Ws[5] = {16, 32, 64, 96, 128};
        for (int wi = 0; wi < 5; wi++) {
            u32 segs = 0, good = 0, good0 = 0;
            for (u32 sgs = 1024; sgs + 1024 < n; sgs += 1024) {
                u32 t = sgs; while (!is_base[t]) t++;               // the true entry
                u32 p = sgs - Ws[wi]; while (p < sgs) { u32 nx; STEP(p, nx); p = nx; }
                segs++; good += p == t; good0 += sgs == t;
            }
            printf("spec walk started %3d positions early: entry right for %.1f%% of the segments (started at the segment start: %.1f%%)\n", Ws[wi], 100.0 * good / segs, 100.0 * good0 / segs);
        }
        free(is_base);
    }
    return 0;
}
