// C ABI of libmtscomp_hip.so (include/mtscomp_hip.h): per-device engine, workspace management, the
// stage pipelines and the host-buffer entry points.  No CPU fallback anywhere: without a gfx950 device
// every compute entry point returns MTS_E_NODEV.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <future>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

#include "common.h"

namespace mts {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// (device, kernel) pairs whose dynamic-LDS limit has been raised
int ensure_dynamic_lds(const void *kernel, int bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<int, const void *>> done;
    int dev = -1;
    MTS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (auto &d : done) if (d.first == dev && d.second == kernel) return MTS_OK;
    MTS_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.push_back({dev, kernel});
    return MTS_OK;
}

static const LevelCfg LEVELS[10] = {{0, 0, 0, 0},     {4, 4, 8, 4},       {4, 5, 16, 8},       {4, 6, 32, 32},
                                    {4, 4, 16, 16},   {8, 16, 32, 32},    {8, 16, 128, 128},   {8, 32, 128, 256},
                                    {32, 128, 258, 1024}, {32, 258, 258, 4096}};

void drop_device_caches();      // frees the decoded-chunk caches of the current device (called when a workspace allocation fails)

// grow-only device buffer
// MTS_ARENA_GB=N (experiment, round 6; default off): the workspaces of a device come out of ONE allocation of N GiB made when
// the first of them is asked for, 2 MiB-aligned pieces handed out one behind the other (a buffer that grows takes a new piece;
// the arena is given back by mts_release).  Asks whether k_match5's three times -- the physical placement of its workspace,
// tools/m5_addr_times.py -- go away when the placement is one big block instead of a dozen allocations made between others.
struct Arena { u8 *base = nullptr; size_t cap = 0, used = 0; bool tried = false; };
static Arena g_arena[64];
static size_t arena_gb() { static const size_t v = [] { const char *e = getenv("MTS_ARENA_GB"); return e ? (size_t)atoll(e) : (size_t)0; }(); return v; }
static void *arena_take(size_t bytes)
{
    if (!arena_gb()) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    Arena &A = g_arena[dev];
    if (!A.tried) {
        A.tried = true;
        if (hipMalloc((void **)&A.base, arena_gb() << 30) == hipSuccess) A.cap = arena_gb() << 30; else { (void)hipGetLastError(); A.base = nullptr; }
    }
    const size_t need = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (!A.base || A.used + need > A.cap) return nullptr;
    void *p = A.base + A.used;
    A.used += need;
    return p;
}
static void arena_reset()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    Arena &A = g_arena[dev];
    if (A.base) (void)hipFree(A.base);
    A = Arena();
}

struct DBuf {
    void *p = nullptr;
    size_t cap = 0;
    u64 gen = 0;              // bumped whenever the buffer is (re)allocated, freed or an allocation fails: what it held is gone
    bool in_arena = false;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return MTS_OK;
        gen++;
        if (p && !in_arena) (void)hipFree(p);
        p = nullptr; cap = 0; in_arena = false;
        const size_t want = bytes + bytes / 8 + 4096;
        if (void *a = arena_take(want)) { p = a; cap = want; in_arena = true; return MTS_OK; }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, bytes);
            if (e != hipSuccess) { (void)hipGetLastError(); drop_device_caches(); e = hipMalloc(&p, bytes); }      // decoded chunks are only a cache
            if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return MTS_E_NOMEM; }
            cap = bytes;
        } else cap = want;
        return MTS_OK;
    }
    void release() { if (p && !in_arena) (void)hipFree(p); p = nullptr; cap = 0; in_arena = false; gen++; }
    template <typename T> T *as() { return (T *)p; }
};

constexpr int MAX_STAGES = 24;

struct Engine {
    int dev = -1;
    std::mutex mu;
    hipStream_t own = nullptr;
    // compress workspace
    DBuf stream, sort_a, sort_b, sort_ws, tables, tokens, marks, segbuf, blk, blkcodes, blkhdr, desc, adler, misc;
    DBuf fast_lists, fast_state;         // levels 1..3: candidate lists of two phases, per-chunk state of the in-order walk
    hipStream_t fast_st = nullptr;       // ... and the stream the lists are made on, with its events (lists ready x2, lists read x2, inputs ready)
    hipEvent_t fast_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // host-API staging
    DBuf h_in, h_out;
    // host copies of a batch's descriptors on their way to the device: they live here until the next batch replaces them, i.e.
    // past the hipStreamSynchronize that ends the batch they belong to (hipMemcpyAsync from pageable memory is not promised to
    // have read its source when it returns)
    std::vector<u8> host_stage[2];
    // pinned pieces the host entry points move user memory through (pageable memory crosses the bus at a fraction of the
    // link's rate, and a fresh destination array takes its page faults on the copying thread): two pieces, so that the
    // DMA of one overlaps the host threads copying the other
    // (round 6: one set per direction -- the host entry points copy the next piece in and the piece before out on two host
    //  threads while the device works on the current one)
    struct Stager {
        void *pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        std::mutex mu;                                   // (a stager's pieces belong to one copy at a time)
        void release()
        {
            for (int k = 0; k < 2; k++) { if (pin[k]) (void)hipHostFree(pin[k]); pin[k] = nullptr; if (ev[k]) (void)hipEventDestroy(ev[k]); ev[k] = nullptr; }
            if (st) (void)hipStreamDestroy(st);
            st = nullptr;
        }
    } stg[2];                                            // [0]: host -> device, [1]: device -> host
    // inflate workspace
    DBuf inf_scratch, inf_desc, segsums;
    // geometry of the last compress batch whose per-segment / per-block / per-tile descriptors are on the device (a recording is
    // compressed batch after batch of the same shape: the 10 MB of index arrays need not be rebuilt and copied every call)
    // (valid while the three buffers are the allocations the arrays were copied into: DBuf::gen, not the address -- a buffer
    // freed by mts_release() and allocated again usually comes back at the same address with nothing in it)
    std::vector<u32> geo_n;
    u64 geo_seg = ~0ull, geo_blk = ~0ull, geo_desc = ~0ull;
    // stage timing
    hipEvent_t ev[MAX_STAGES + 1];
    bool ev_ok = false;
    const char *stage_name[MAX_STAGES];
    int n_stage = 0;
    float stage_ms[MAX_STAGES];
    int n_stage_done = 0;
    const char *done_name[MAX_STAGES];

    int init_events()
    {
        if (ev_ok) return MTS_OK;
        for (int i = 0; i <= MAX_STAGES; i++) MTS_HIP(hipEventCreate(&ev[i]));
        ev_ok = true;
        return MTS_OK;
    }
    int init_fast_streams()
    {
        if (fast_st) return MTS_OK;
        MTS_HIP(hipStreamCreateWithFlags(&fast_st, hipStreamNonBlocking));
        for (auto &e : fast_ev) MTS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return MTS_OK;
    }
    void t_begin(hipStream_t st) { n_stage = 0; (void)hipEventRecord(ev[0], st); }
    void t_mark(hipStream_t st, const char *name)
    {
        if (n_stage < MAX_STAGES) { stage_name[n_stage] = name; n_stage++; (void)hipEventRecord(ev[n_stage], st); }
    }
    void t_collect(bool accumulate)
    {
        if (!accumulate) { n_stage_done = 0; }
        for (int i = 0; i < n_stage; i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            int k = -1;
            for (int j = 0; j < n_stage_done; j++) if (!strcmp(done_name[j], stage_name[i])) k = j;
            if (k < 0 && n_stage_done < MAX_STAGES) { k = n_stage_done++; done_name[k] = stage_name[i]; stage_ms[k] = 0; }
            if (k >= 0) stage_ms[k] += ms;
        }
    }
    void release_all()
    {
        DBuf *all[] = {&stream, &sort_a, &sort_b, &sort_ws, &tables, &tokens, &marks, &segbuf, &blk, &blkcodes, &blkhdr, &desc,
                       &adler, &misc, &h_in, &h_out, &inf_scratch, &inf_desc, &segsums, &fast_lists, &fast_state};
        for (DBuf *b : all) b->release();
        arena_reset();                                   // (every piece of it has just been let go)
        geo_n.clear();
        for (auto &g : stg) g.release();
        if (fast_st) (void)hipStreamDestroy(fast_st);
        fast_st = nullptr;
        for (auto &e : fast_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    }
};

static std::mutex g_mu;
static std::vector<Engine *> g_engines;
static int g_ndev = -2;

static int device_count()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ndev != -2) return g_ndev;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    int ok = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) break;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) break;     // code objects are gfx950 only
        ok++;
    }
    g_ndev = ok;
    g_engines.assign(ok, nullptr);
    return g_ndev;
}

static int get_engine(int device, Engine **out)
{
    const int n = device_count();
    if (n <= 0) { set_error("no gfx950 device visible (libmtscomp_hip has no CPU path)"); return MTS_E_NODEV; }
    if (device < 0 || device >= n) { set_error("device %d out of range (%d visible)", device, n); return MTS_E_ARG; }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_engines[device]) { g_engines[device] = new Engine(); g_engines[device]->dev = device; }
    *out = g_engines[device];
    return MTS_OK;
}

constexpr int PARSE_PARALLEL_ROUNDS = 96;     // parallel correction rounds of the speculative parse (~40 us each) before the in-order pass:
                                              // chains of a few dozen segments (a dead channel) are cheaper in parallel, whole-chunk chains are not

static long compress_bound(long n) { return n + (n >> 12) + (n >> 14) + (n >> 25) + 13; }

// ------------------------------------------------------------------------------------------------
// compress pipeline over one sub-batch of chunks (device resident)
// ------------------------------------------------------------------------------------------------
struct DebugTap {              // optional host copies of intermediates (tests)
    unsigned *t_full = nullptr, *t_quarter = nullptr;
    unsigned short *tokens = nullptr;
    long *n_tokens = nullptr;
};

static int compress_batch(Engine &E, hipStream_t st, const u8 *d_raw, bool raw_is_stream, int nc, int sz,
                          const long *bounds, int n_chunks, int flags, int level, u8 *d_out, const long *slot_off,
                          long *out_sizes, bool accumulate_times, DebugTap *tap)
{
    const LevelCfg cfg = LEVELS[level];
    std::vector<ChunkDesc> cd(n_chunks);
    std::vector<TileDesc> tiles;
    u64 soff = 0, toff = 0, sorted_off = 0;
    u32 nseg = 0, nblk = 0, max_rows = 0;
    const u64 row_bytes = (u64)nc * sz;
    for (int i = 0; i < n_chunks; i++) {
        const u64 rows = (u64)(bounds[i + 1] - bounds[i]);
        const u64 n = raw_is_stream ? rows : rows * row_bytes;
        if (n >= (1ull << 31)) { set_error("chunk %d is %llu bytes; chunks must be < 2 GiB", i, (unsigned long long)n); return MTS_E_ARG; }
        ChunkDesc &c = cd[i];
        c.stream_off = soff; c.tok_off = toff; c.out_off = (u64)slot_off[i];
        c.raw_off = (u64)(bounds[i] - bounds[0]) * (raw_is_stream ? 1 : row_bytes);
        c.n = (u32)n; c.n_rows = (u32)rows;
        c.seg0 = nseg; c.nseg = (u32)((n + SEG - 1) / SEG);
        c.blk0 = nblk; c.blk_cap = (u32)(n / BLOCK_TOKENS + 2);
        if (c.out_off & 15) { set_error("output slot %d is not 16-byte aligned", i); return MTS_E_ARG; }
        nseg += c.nseg; nblk += c.blk_cap;
        if (rows > max_rows) max_rows = (u32)rows;
        c.tile0 = (u32)tiles.size(); c.pad = 0;
        for (u64 a = 0; a < n; a += TILE) {
            TileDesc t;
            t.stream_off = soff; t.sorted_off = sorted_off; t.n = (u32)n; t.a = (u32)a;
            t.w = (u32)(a >= (u64)HALO ? a - HALO : 0);
            t.own_end = (u32)(a + TILE < n ? a + TILE : n);
            const u64 hashed_end = n >= 3 ? (t.own_end < n - 2 ? t.own_end : n - 2) : 0;
            t.wlen = hashed_end > t.w ? (u32)(hashed_end - t.w) : 0;
            t.chunk = (u32)i;
            sorted_off += align_up(t.wlen, 64);
            tiles.push_back(t);
        }
        soff += align_up(n + STREAM_PAD, STREAM_ALIGN);
        toff += n + 1;
    }
    const u64 stream_bytes = soff + STREAM_PAD;
    // ---- workspace ----
    int rc;
    if (!raw_is_stream) { if ((rc = E.stream.ensure(stream_bytes))) return rc; }
    const size_t sort_n = align_up(sorted_off + 64, 64);          // 32-bit keys
    // (the first pass's keys are dead once the sort is done: the parse keeps its marks there)
    const size_t marks_words = parse_marks_words((size_t)nseg + 64);
    if ((rc = E.sort_a.ensure((sort_n > marks_words ? sort_n : marks_words) * 4))) return rc;
    if ((rc = E.sort_b.ensure(sort_n * 4))) return rc;
    if ((rc = E.sort_ws.ensure(hash_sort_ws_bytes((int)tiles.size())))) return rc;
    // one word per position (levels 1..3: the inverse map) + for levels 4..9 the side table of the quarter-budget results, which is
    // written and read at a fraction of a percent of the positions only
    const size_t table_words = align_up(stream_bytes + 64, 64);
    if ((rc = E.tables.ensure(table_words * 4 * (level < 4 ? 1 : 2)))) return rc;
    if ((rc = E.tokens.ensure((toff + 64) * 4))) return rc;
    if ((rc = E.segbuf.ensure((size_t)(nseg + 64) * 4 * (7 + parse_cp_words()) + 256))) return rc;
    if ((rc = E.blk.ensure((size_t)(nblk + 1) * (sizeof(BlockRec) + 8) + 256))) return rc;
    if ((rc = E.blkcodes.ensure((size_t)(nblk + 1) * BLK_CODE_WORDS * 4))) return rc;
    if ((rc = E.blkhdr.ensure((size_t)(nblk + 1) * BLK_HDR_WORDS * 4))) return rc;
    const size_t desc_bytes = align_up(sizeof(ChunkDesc) * n_chunks, 256) + align_up(sizeof(TileDesc) * (tiles.size() + 1), 256) +
                              align_up(sizeof(ChunkOut) * n_chunks, 256);
    if ((rc = E.desc.ensure(desc_bytes))) return rc;
    if ((rc = E.adler.ensure(sizeof(u64) * 2 * n_chunks + 256 + MATCH_SINK_BYTES))) return rc;      // + the flag words and the match stage's sink behind them
    if ((rc = E.init_events())) return rc;

    u8 *dp = E.desc.as<u8>();
    ChunkDesc *d_chunks = (ChunkDesc *)dp; dp += align_up(sizeof(ChunkDesc) * n_chunks, 256);
    TileDesc *d_tiles = (TileDesc *)dp; dp += align_up(sizeof(TileDesc) * (tiles.size() + 1), 256);
    ChunkOut *d_cout = (ChunkOut *)dp;
    // per-segment arrays
    u32 *sg = E.segbuf.as<u32>();
    ParseBufs pb;
    const size_t SN = nseg + 64;
    pb.entry = sg; pb.exit_a = sg + SN; pb.exit_b = sg + 2 * SN; pb.cnt = sg + 3 * SN; pb.tokbase = sg + 4 * SN;
    pb.seg_chunk = sg + 5 * SN; pb.seg_start = sg + 6 * SN;
    pb.cp = sg + 7 * SN;
    pb.changed = (int *)(E.adler.as<u8>() + sizeof(u64) * 2 * n_chunks);
    pb.marks = E.sort_a.as<u32>();
    // block arrays
    BlockRec *d_blocks = E.blk.as<BlockRec>();
    u32 *d_blk_chunk = (u32 *)(d_blocks + nblk + 1);
    u32 *d_blk_in_start = d_blk_chunk + nblk + 1;

    MTS_HIP(hipMemcpyAsync(d_chunks, cd.data(), sizeof(ChunkDesc) * n_chunks, hipMemcpyHostToDevice, st));
    {
        // the index arrays depend on the chunk sizes only: kept on the device while the next batch has the same sizes and
        // the buffers have not moved
        std::vector<u32> sizes_now(n_chunks);
        for (int i = 0; i < n_chunks; i++) sizes_now[i] = cd[i].n;
        const bool same = sizes_now == E.geo_n && E.geo_seg == E.segbuf.gen && E.geo_blk == E.blk.gen && E.geo_desc == E.desc.gen;
        if (!same) {
            E.geo_n.clear();                                      // (not valid again until everything below is on its way)
            std::vector<u32> h_seg(2 * (size_t)nseg), h_blk_chunk(nblk + 1);
            for (int i = 0; i < n_chunks; i++) {
                for (u32 k = 0; k < cd[i].nseg; k++) { h_seg[cd[i].seg0 + k] = (u32)i; h_seg[nseg + cd[i].seg0 + k] = k * SEG; }
                for (u32 k = 0; k < cd[i].blk_cap; k++) h_blk_chunk[cd[i].blk0 + k] = (u32)i;
            }
            if (!tiles.empty()) MTS_HIP(hipMemcpyAsync(d_tiles, tiles.data(), sizeof(TileDesc) * tiles.size(), hipMemcpyHostToDevice, st));
            if (nseg) {
                MTS_HIP(hipMemcpyAsync(pb.seg_chunk, h_seg.data(), 4 * (size_t)nseg, hipMemcpyHostToDevice, st));
                MTS_HIP(hipMemcpyAsync(pb.seg_start, h_seg.data() + nseg, 4 * (size_t)nseg, hipMemcpyHostToDevice, st));
            }
            MTS_HIP(hipMemcpyAsync(d_blk_chunk, h_blk_chunk.data(), 4 * (size_t)nblk, hipMemcpyHostToDevice, st));
            // (pageable copies: staged before hipMemcpyAsync returns, so the vectors may go)
            E.geo_n = sizes_now; E.geo_seg = E.segbuf.gen; E.geo_blk = E.blk.gen; E.geo_desc = E.desc.gen;
        }
    }
    MTS_HIP(hipMemsetAsync(d_cout, 0, sizeof(ChunkOut) * n_chunks, st));
    MTS_HIP(hipMemsetAsync(pb.changed, 0, 8, st));           // + the match stage's flag word behind it
    // the host copies above must be complete before the std::vectors go away; they are pageable
    // copies, which hipMemcpyAsync finishes staging before returning.

    E.t_begin(st);
    const u8 *d_stream;
    u64 *d_adler = E.adler.as<u64>();
    std::vector<u64> so(n_chunks); std::vector<u32> nn(n_chunks);
    if (raw_is_stream) {
        // debug path: the caller's bytes already are the transformed stream (one chunk)
        d_stream = d_raw;
        for (int i = 0; i < n_chunks; i++) { so[i] = cd[i].stream_off; nn[i] = cd[i].n; }
        if ((rc = E.misc.ensure(12 * (size_t)n_chunks + 64))) return rc;
        u64 *d_so = E.misc.as<u64>(); u32 *d_nn = (u32 *)(d_so + n_chunks);
        MTS_HIP(hipMemcpyAsync(d_so, so.data(), 8 * (size_t)n_chunks, hipMemcpyHostToDevice, st));
        MTS_HIP(hipMemcpyAsync(d_nn, nn.data(), 4 * (size_t)n_chunks, hipMemcpyHostToDevice, st));
        u32 max_n = 0; for (int i = 0; i < n_chunks; i++) if (cd[i].n > max_n) max_n = cd[i].n;
        if ((rc = launch_adler_stream(st, d_stream, d_so, d_nn, n_chunks, max_n, d_adler, nullptr, 0))) return rc;
    } else {
        if ((rc = launch_delta_transpose(st, d_raw, E.stream.p, d_chunks, n_chunks, max_rows, nc, sz, flags, d_adler))) return rc;
        d_stream = E.stream.as<u8>();
    }
    E.t_mark(st, "delta_transpose");
    u32 *tmp_k = E.sort_a.as<u32>(), *srt_k = E.sort_b.as<u32>();
    u32 *d_tables = E.tables.as<u32>(), *d_quarter = d_tables + table_words;
    u32 *d_flags = (u32 *)(pb.changed + 1);                   // [0] bit 0: the match stage found a hash run out of position order
    u32 *d_tokens = E.tokens.as<u32>();
    int force_ballot = getenv("MTS_SORT_INJECT_DISORDER") ? 2 : 0;      // (test hook: the first sort of the call is deliberately mis-ranked)
    const bool fast = level < 4;                              // deflate_fast: no candidate tables, the walk itself searches (deflate.hip, section F)
    u32 *d_inv = (u32 *)d_tables;                             // levels 1..3: the inverse map lives where the other levels keep the candidate tables
    int fix_rounds = 0;                                       // parallel fix rounds of the parse that counted (they say where the exits are)
    for (;;) {
        if ((rc = launch_hash_sort(st, d_stream, d_tiles, (int)tiles.size(), tmp_k, srt_k, force_ballot, E.sort_ws.p))) return rc;
        E.t_mark(st, force_ballot == 1 ? "hash_sort_retry" : "hash_sort");
        int round = 0;
        bool resort = false;
        if (fast) {
            // one in-order pass per chunk over candidate lists made a phase (W positions of every chunk) at a time
            if ((rc = launch_inverse_map(st, d_stream, d_tiles, (int)tiles.size(), srt_k, d_inv, d_flags))) return rc;
            E.t_mark(st, "inverse_map");
            u32 max_n = 0;
            for (int i = 0; i < n_chunks; i++) if (cd[i].n > max_n) max_n = cd[i].n;
            const u64 K = (u64)fast_list_rows(level);
            const char *be = getenv("MTS_FAST_LIST_BYTES");      // (tests: a tiny budget = many phases)
            const u64 budget = be ? strtoull(be, nullptr, 10) : (u64)8 << 30;
            // two list buffers: the lists of phase k + 1 are made on a second stream while phase k is walked (the walk keeps
            // one wave per chunk busy, the rest of the device is free)
            u64 W = budget / 2 / ((u64)n_chunks * K * 4) / 256 * 256;
            if (W < 256) W = 256;
            if (W > align_up(max_n, 256)) W = align_up(max_n, 256);
            if (W > 1024 && W * 8 > max_n) W = align_up((max_n + 7) / 8, 256);      // at least 8 phases: only the first lists are waited for
            if (max_n) {
                const size_t list_words = (size_t)n_chunks * W * K;
                if ((rc = E.fast_lists.ensure(2 * list_words * 4))) return rc;
                if ((rc = E.fast_state.ensure(fast_seq_state_bytes(n_chunks)))) return rc;
                if ((rc = E.init_fast_streams())) return rc;
                u32 *lists[2] = {E.fast_lists.as<u32>(), E.fast_lists.as<u32>() + list_words};
                MTS_HIP(hipEventRecord(E.fast_ev[4], st));       // the sort and the inverse map
                MTS_HIP(hipStreamWaitEvent(E.fast_st, E.fast_ev[4], 0));
                for (u64 ph = 0; ph * W < max_n; ph++) {
                    const int b = (int)(ph & 1);
                    if (ph >= 2) MTS_HIP(hipStreamWaitEvent(E.fast_st, E.fast_ev[2 + b], 0));      // the walk that read this buffer
                    if ((rc = launch_fast_cands(E.fast_st, d_stream, d_chunks, d_tiles, srt_k, d_inv, lists[b], (u32)W, (u32)ph, n_chunks, level, cfg))) return rc;
                    MTS_HIP(hipEventRecord(E.fast_ev[b], E.fast_st));
                    MTS_HIP(hipStreamWaitEvent(st, E.fast_ev[b], 0));
                    if ((rc = launch_fast_seq(st, d_stream, d_chunks, d_tiles, srt_k, d_inv, lists[b], (u32)W, (u32)ph, E.fast_state.p, n_chunks, level, cfg,
                                              d_tokens, d_blk_in_start, d_cout))) return rc;
                    MTS_HIP(hipEventRecord(E.fast_ev[2 + b], st));
                }
            }
            int hflags[2] = {0, 0};                              // {-, sort-order flag}
            MTS_HIP(hipMemcpyAsync(hflags, pb.changed, 8, hipMemcpyDeviceToHost, st));
            MTS_HIP(hipStreamSynchronize(st));
            if (hflags[1] & 1) resort = true;
        } else {
        if (getenv("MTS_DEBUG_ADDR"))                                 // (tools/m5_addr_times.py: does the match stage's time follow where its buffers lie?)
            fprintf(stderr, "[addr] stream %p sorted %p tables %p quarter %p tiles %zu\n", (void *)d_stream, (void *)srt_k, (void *)d_tables, (void *)d_quarter, tiles.size());
        if ((rc = launch_match(st, d_stream, d_tiles, (int)tiles.size(), srt_k, d_tables, d_quarter, cfg, d_flags, tap && tap->t_full ? 1 : 0))) return rc;
        E.t_mark(st, "match");
        u32 max_nseg = 0;
        for (int i = 0; i < n_chunks; i++) if (cd[i].nseg > max_nseg) max_nseg = cd[i].nseg;
        if ((rc = launch_parse_spec(st, d_tables, d_quarter, d_chunks, pb, (int)nseg, cfg, n_chunks, max_nseg))) return rc;
        for (;;) {
            if ((rc = launch_parse_fix(st, d_tables, d_quarter, d_chunks, pb, (int)nseg, cfg, round))) return rc;
            round++;
            int hflags[2] = {0, 0};                              // {changed, match-stage flags}
            MTS_HIP(hipMemcpyAsync(hflags, pb.changed, 8, hipMemcpyDeviceToHost, st));
            MTS_HIP(hipStreamSynchronize(st));
            static const bool debug_flags = getenv("MTS_DEBUG_FLAGS") != nullptr;      // (read once)
            if (debug_flags) fprintf(stderr, "[flags] level %d round %d changed %d match-flags %d force_ballot %d\n", level, round, hflags[0], hflags[1], force_ballot);
            if (hflags[1] & 1) { resort = true; break; }
            if (!hflags[0]) break;
            MTS_HIP(hipMemsetAsync(pb.changed, 0, 4, st));
            if (round >= PARSE_PARALLEL_ROUNDS) {            // (runs, periodic data: the parse does not re-synchronise) the rest in order
                if ((rc = launch_parse_fix_serial(st, d_tables, d_quarter, d_chunks, pb, n_chunks, cfg, round))) return rc;
                break;
            }
        }
        }
        fix_rounds = round;
        if (!resort) break;
        // The lane-ordered LDS ranking of the sort (deflate.hip: rank_pass) did not hold: byte identity with zlib needs
        // position-ordered chains, so the stage is repeated with the ballot ranking, which relies on nothing.
        if (force_ballot == 1) { set_error("hash sort: runs out of position order even with the ballot ranking"); return MTS_E_INTERNAL; }
        force_ballot = 1;
        MTS_HIP(hipMemsetAsync(pb.changed, 0, 8, st));
    }
    // after an odd number of fix rounds the current exits live in exit_b; nothing downstream needs them
    E.t_mark(st, fast ? "fast_walk" : "parse_fixpoint");
    if (!fast) {                                              // (levels 1..3: the in-order walk has written tokens and counts)
        if ((rc = launch_parse_count(st, d_tables, d_chunks, pb, (int)nseg, n_chunks, cfg, d_cout))) return rc;
        u32 max_nseg = 0;
        for (int i = 0; i < n_chunks; i++) if (cd[i].nseg > max_nseg) max_nseg = cd[i].nseg;
        if ((rc = launch_parse_emit_marks(st, d_stream, d_tables, d_quarter, d_chunks, pb, fix_rounds, d_tokens, d_blk_in_start, d_cout, n_chunks, max_nseg))) return rc;
    }
    E.t_mark(st, "parse_emit");
    if ((rc = launch_block_trees(st, d_chunks, d_blk_chunk, (int)nblk, d_tokens, d_blk_in_start, d_cout, d_blocks,
                                 E.blkcodes.as<u32>(), E.blkhdr.as<u32>(), fast ? 1 : 0))) return rc;
    if ((rc = launch_block_layout(st, d_chunks, n_chunks, d_blocks, d_cout, d_adler))) return rc;
    if ((rc = launch_zero_edges(st, d_chunks, d_blk_chunk, (int)nblk, d_blocks, d_cout, d_out))) return rc;      // (the words the packer ORs into)
    E.t_mark(st, "block_trees");
    if ((rc = launch_block_pack(st, d_stream, d_chunks, d_blk_chunk, (int)nblk, d_tokens, d_blocks, E.blkcodes.as<u32>(),
                                E.blkhdr.as<u32>(), d_cout, d_out, level))) return rc;
    E.t_mark(st, "block_pack");
    std::vector<ChunkOut> h_cout(n_chunks);
    MTS_HIP(hipMemcpyAsync(h_cout.data(), d_cout, sizeof(ChunkOut) * n_chunks, hipMemcpyDeviceToHost, st));
    MTS_HIP(hipStreamSynchronize(st));
    E.t_collect(accumulate_times);
    for (int i = 0; i < n_chunks; i++) out_sizes[i] = (long)h_cout[i].nbytes;
    if (tap) {
        // single-chunk debug taps
        const u32 n = cd[0].n;
        if (tap->t_full && n) {
            // (the tap had the match stage write the side table for every position)
            std::vector<u32> hf(n), hq(n);
            MTS_HIP(hipMemcpy(hf.data(), d_tables + cd[0].stream_off, 4 * (size_t)n, hipMemcpyDeviceToHost));
            MTS_HIP(hipMemcpy(hq.data(), d_quarter + cd[0].stream_off, 4 * (size_t)n, hipMemcpyDeviceToHost));
            auto unpack = [](u32 e) -> unsigned { return (e & 0x7fffu) ? ((((e >> 15) & 0xffu) + MIN_MATCH) << 16) | (e & 0x7fffu) : 0u; };
            for (u32 i = 0; i < n; i++) {
                tap->t_full[i] = unpack(hf[i]);
                tap->t_quarter[i] = unpack(hq[i]);
                // the flags must say what the two results say
                const bool differs = hq[i] != (hf[i] & 0x7fffffu);
                const u32 fl = hf[i] >> 23;
                if ((fl != 0) != differs || (fl == 2) != (differs && (tap->t_quarter[i] >> 16) > (u32)cfg.good)) {
                    set_error("table entry %u: flags %u do not describe full %08x / quarter %08x", i, fl, hf[i], hq[i]);
                    return MTS_E_INTERNAL;
                }
            }
        }
        if (tap->tokens) {
            const u32 nt = h_cout[0].ntok;
            if (nt) MTS_HIP(hipMemcpy(tap->tokens, d_tokens + cd[0].tok_off, 4 * (size_t)nt, hipMemcpyDeviceToHost));
            *tap->n_tokens = nt;
        }
    }
    return MTS_OK;
}

// ------------------------------------------------------------------------------------------------
// user memory <-> device through pinned pieces
// ------------------------------------------------------------------------------------------------
constexpr size_t PIN_PIECE = (size_t)32 << 20;

static int host_threads()
{
    static const int n = [] { const char *e = getenv("MTS_HOST_THREADS"); int v = e ? atoi(e) : 8; return v < 1 ? 1 : v > 64 ? 64 : v; }();
    return n;
}

// memcpy by several threads (a fresh destination is faulted in by all of them at once).  The threads are a pool that lives as
// long as the library (round 6): started per call they cost ~100 us per copy -- nothing against a 32 MiB piece, a third of the
// time of the 4 MiB pieces a cold Reader window is moved in.  One copy at a time uses the pool; a second caller (the other
// direction of a pipelined host call) copies on its own thread instead of waiting.
namespace {
struct CopyPool {
    std::mutex mu, use;                                 // mu: the job; use: one parallel copy at a time
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> th;
    u8 *dst = nullptr; const u8 *src = nullptr; size_t n = 0, per = 0;
    u64 gen = 0; int pending = 0; bool stop = false;
    void worker(int t, u64 seen /* the job counter when the thread was made: jobs published before are not its */)
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            const size_t a = (size_t)t * per;
            u8 *d = dst; const u8 *s = src; const size_t nn = n, pp = per;
            lk.unlock();
            if (a < nn) memcpy(d + a, s + a, nn - a < pp ? nn - a : pp);
            lk.lock();
            if (--pending == 0) cv_done.notify_one();
        }
    }
    void run(void *d, const void *s, size_t bytes, int nt)
    {
        if ((int)th.size() + 1 < nt) {
            std::lock_guard<std::mutex> lk(mu);
            for (int t = (int)th.size() + 1; t < nt; t++) th.emplace_back(&CopyPool::worker, this, t, gen);
        }
        // (the share is rounded UP before it is aligned: with n / nt an exact multiple of 4096 and n % nt != 0 the threads' shares
        //  ended n % nt bytes short of n -- the last bytes of such a copy were never made; found by tools/fuzz_gpu.py, seed 301)
        const size_t p = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> lk(mu);
            dst = (u8 *)d; src = (const u8 *)s; n = bytes; per = p; pending = (int)th.size(); gen++;
        }
        cv_go.notify_all();
        memcpy(d, s, bytes < p ? bytes : p);                            // share 0 on the calling thread
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};
CopyPool g_copy_pool;
}  // namespace
static void par_memcpy(void *dst, const void *src, size_t n)
{
    const int nt = n < ((size_t)1 << 20) ? 1 : host_threads();
    if (nt == 1) { memcpy(dst, src, n); return; }
    std::unique_lock<std::mutex> one(g_copy_pool.use, std::try_to_lock);
    if (!one.owns_lock()) { memcpy(dst, src, n); return; }            // (the pool is busy with the other direction's copy)
    g_copy_pool.run(dst, src, n, nt);
}

static int pin_init(Engine::Stager &G)
{
    if (G.pin[0]) return MTS_OK;
    for (int k = 0; k < 2; k++) {
        if (hipHostMalloc(&G.pin[k], PIN_PIECE, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); G.pin[k] = nullptr; }
        if (G.pin[k] && hipEventCreateWithFlags(&G.ev[k], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(G.pin[k]); G.pin[k] = nullptr; }
    }
    if (G.pin[0] && G.pin[1] && hipStreamCreateWithFlags(&G.st, hipStreamNonBlocking) == hipSuccess) return MTS_OK;
    (void)hipGetLastError();
    for (int k = 0; k < 2; k++) { if (G.pin[k]) (void)hipHostFree(G.pin[k]); G.pin[k] = nullptr; }
    return MTS_E_NOMEM;           // (the callers fall back to plain copies)
}

// device -> user memory, any number of pieces (dst, src, bytes).  The device data must be complete (the caller synchronised the
// stream that produced it).  The DMA of the next piece runs while the host threads copy the one before out of its pinned buffer.
struct CopyItem { void *dst; const void *src; size_t n; };
static bool host_ptr_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int staged_d2h_multi(Engine &E, const std::vector<CopyItem> &segs)
{
    Engine::Stager &G = E.stg[1];
    std::lock_guard<std::mutex> lk(G.mu);
    size_t total = 0;
    for (auto &s : segs) total += s.n;
    // a destination that is pinned already (mts_host_alloc) takes the DMA itself: no pinned piece in between, no host copy
    if (total >= ((size_t)1 << 20) && !segs.empty()) {
        bool all_pinned = true;
        for (auto &s : segs) if (s.n && !host_ptr_pinned(s.dst)) { all_pinned = false; break; }
        if (all_pinned) {
            hipStream_t cs = pin_init(G) == MTS_OK ? G.st : nullptr;
            for (auto &s : segs) if (s.n) MTS_HIP(hipMemcpyAsync(s.dst, s.src, s.n, hipMemcpyDeviceToHost, cs));
            MTS_HIP(hipStreamSynchronize(cs));
            return MTS_OK;
        }
    }
    if (total < ((size_t)1 << 20) || pin_init(G) != MTS_OK) {
        for (auto &s : segs) if (s.n) MTS_HIP(hipMemcpy(s.dst, s.src, s.n, hipMemcpyDeviceToHost));
        return MTS_OK;
    }
    std::vector<CopyItem> items;                      // cut to pinned-piece size
    for (auto &s : segs)
        for (size_t o = 0; o < s.n; o += PIN_PIECE) items.push_back({(u8 *)s.dst + o, (const u8 *)s.src + o, s.n - o < PIN_PIECE ? s.n - o : PIN_PIECE});
    auto issue = [&](size_t k) -> int {
        MTS_HIP(hipMemcpyAsync(G.pin[k & 1], items[k].src, items[k].n, hipMemcpyDeviceToHost, G.st));
        MTS_HIP(hipEventRecord(G.ev[k & 1], G.st));
        return MTS_OK;
    };
    int rc;
    if (!items.empty() && (rc = issue(0))) return rc;
    for (size_t k = 0; k < items.size(); k++) {
        if (k + 1 < items.size() && (rc = issue(k + 1))) return rc;
        MTS_HIP(hipEventSynchronize(G.ev[k & 1]));
        par_memcpy(items[k].dst, G.pin[k & 1], items[k].n);
    }
    return MTS_OK;
}
static int staged_d2h(Engine &E, void *dst, const void *d_src, size_t n) { return staged_d2h_multi(E, {{dst, d_src, n}}); }

// user memory -> device; complete on return
static int staged_h2d(Engine &E, void *d_dst, const void *src, size_t n)
{
    Engine::Stager &G = E.stg[0];
    std::lock_guard<std::mutex> lk(G.mu);
    if (n >= ((size_t)1 << 20) && host_ptr_pinned(src)) {            // (a pinned source: the DMA reads it directly; on the stager's own
        hipStream_t cs = pin_init(G) == MTS_OK ? G.st : nullptr;   //  stream, so that it runs beside the kernels of the piece before)
        MTS_HIP(hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, cs));
        MTS_HIP(hipStreamSynchronize(cs));
        return MTS_OK;
    }
    // (small copies take the plain call -- which, on the null stream, waits for the kernels there: the pieces of a pipelined call are
    //  megabytes and go through the stager's own stream)
    if (n < ((size_t)1 << 20) || pin_init(G) != MTS_OK) { MTS_HIP(hipMemcpy(d_dst, src, n, hipMemcpyHostToDevice)); return MTS_OK; }
    // pieces of the page-locked buffers' size -- or, for a transfer of a few MB (the compressed bytes of a cold window), a quarter
    // of it, so that the DMA of one piece runs under the host copy of the next
    size_t piece = PIN_PIECE;
    if (n < 4 * PIN_PIECE) { piece = ((n + 3) / 4 + 4095) & ~(size_t)4095; if (piece < ((size_t)2 << 20)) piece = (size_t)2 << 20; if (piece > PIN_PIECE) piece = PIN_PIECE; }
    const size_t np = (n + piece - 1) / piece;
    auto len = [&](size_t k) { return k + 1 < np ? piece : n - k * piece; };
    for (size_t k = 0; k < np; k++) {
        if (k >= 2) MTS_HIP(hipEventSynchronize(G.ev[k & 1]));       // the DMA out of this piece two rounds ago
        par_memcpy(G.pin[k & 1], (const u8 *)src + k * piece, len(k));
        MTS_HIP(hipMemcpyAsync((u8 *)d_dst + k * piece, G.pin[k & 1], len(k), hipMemcpyHostToDevice, G.st));
        MTS_HIP(hipEventRecord(G.ev[k & 1], G.st));
    }
    MTS_HIP(hipStreamSynchronize(G.st));
    return MTS_OK;
}

// The host entry points work piece by piece (round 6): while the device compresses / inflates piece k, one host thread copies
// piece k + 1 in and another copies the result of piece k - 1 out -- the three used to follow each other (PCIe in, kernels, PCIe
// out: 21 / 31 GB/s for the 60-chunk recording where the kernels alone do 46 / 134).  A piece is MTS_PIPE_BYTES of raw data
// (default 256 MiB; 0 = one piece, the old behaviour): big enough that the kernels lose nothing, small enough that a recording
// of a few hundred MB already overlaps.
// a piece's copy on a helper thread; when no thread can be started (std::system_error) the copy is made at once, on this one
template <class F>
static std::future<int> copy_beside(F &&f, int k)
{
    try {
        return std::async(std::launch::async, f, k);
    } catch (...) {
        std::promise<int> p;
        int rc = MTS_E_INTERNAL;
        try { rc = f(k); } catch (...) {}
        p.set_value(rc);
        return p.get_future();
    }
}
static size_t pipe_piece_bytes()
{
    const char *e = getenv("MTS_PIPE_BYTES");
    return e ? (size_t)atoll(e) : ((size_t)256 << 20);
}
static std::vector<int> pipe_pieces(const long *n_rows_or_bounds, bool is_bounds, int n_chunks, u64 row_bytes)
{
    std::vector<int> pb = {0};
    const size_t piece = pipe_piece_bytes();
    if (piece) {
        u64 acc = 0;
        for (int i = 0; i < n_chunks; i++) {
            const u64 n = (u64)(is_bounds ? n_rows_or_bounds[i + 1] - n_rows_or_bounds[i] : n_rows_or_bounds[i]) * row_bytes;
            if (acc && acc + n > piece) { pb.push_back(i); acc = 0; }
            acc += n;
        }
    }
    pb.push_back(n_chunks);
    return pb;
}

// split a call into sub-batches that fit the workspace budget (stream bytes per sub-batch) and the grid (several kernels
// take the chunk index from blockIdx.y, which ends at 65535)
constexpr int MAX_BATCH_CHUNKS = 32768;
static size_t batch_budget_bytes(bool in_order_walk = false)
{
    const char *e = getenv("MTS_BATCH_BYTES");               // read per call: tests force small sub-batches with it
    // 3 GiB of stream -> ~75 GiB of workspace.  Levels 1..3: 6 GiB (17 bytes of workspace per byte + the candidate lists: ~110
    // GiB) -- their in-order walk takes as long for one chunk as for 256, so fewer, larger sub-batches are what counts there
    size_t v = e ? (size_t)atoll(e) : ((size_t)(in_order_walk ? 6 : 3) << 30);
    if (v < (1u << 20)) v = 1u << 20;
    return v;
}

static int dev_compress(Engine &E, hipStream_t st, const void *d_raw, int nc, int sz, const long *bounds, int n_chunks,
                        int flags, int level, u8 *d_out, const long *slot_off, long *out_sizes, bool add_times = false)
{
    if (level == -1) level = 6;
    if (level < 1 || level > 9) { set_error("level %d out of range", level); return MTS_E_ARG; }
    if (sz != 1 && sz != 2 && sz != 4 && sz != 8) { set_error("itemsize %d unsupported", sz); return MTS_E_ARG; }
    if ((flags & MTS_FLAG_FLOAT) && sz != 4 && sz != 8) { set_error("float items of %d bytes unsupported", sz); return MTS_E_ARG; }
    if (nc <= 0 || n_chunks < 0) return MTS_E_ARG;
    MTS_HIP(hipSetDevice(E.dev));
    const size_t budget = batch_budget_bytes(level < 4);
    const u64 row_bytes = (u64)nc * sz;
    int i = 0;
    bool first = !add_times;
    while (i < n_chunks) {
        int j = i;
        size_t acc = 0;
        while (j < n_chunks) {
            const size_t n = (size_t)(bounds[j + 1] - bounds[j]) * row_bytes;
            if (j > i && (acc + n > budget || j - i >= MAX_BATCH_CHUNKS)) break;
            acc += n; j++;
        }
        const u8 *raw = (const u8 *)d_raw + (u64)(bounds[i] - bounds[0]) * row_bytes;
        int rc = compress_batch(E, st, raw, false, nc, sz, bounds + i, j - i, flags, level, d_out, slot_off + i, out_sizes + i,
                                !first, nullptr);
        if (rc) return rc;
        first = false;
        i = j;
    }
    if (n_chunks == 0) E.n_stage_done = 0;
    return MTS_OK;
}


// ------------------------------------------------------------------------------------------------
// decompress pipeline over one sub-batch (device resident)
// ------------------------------------------------------------------------------------------------
static int decompress_batch(Engine &E, hipStream_t st, const u8 *d_cdata, const long *c_off, const long *c_len,
                            const long *n_rows, int n_chunks, int nc, int sz, int flags, u8 *d_out, const long *out_off,
                            int *status, int times /* 0: these stages replace the recorded ones, 1: are added, 2: are not recorded */,
                            u8 *stream_copy_host /* debug: first chunk's stream */,
                            int nc_full = 0 /* > nc: the chunks have nc_full channels and only the first nc are decoded */,
                            bool size_verdict = true /* a chunk of another size than expected gets its check value looked at */)
{
    if (nc_full <= nc) nc_full = 0;
    std::vector<InfChunk> ic(n_chunks);
    std::vector<u64> so(n_chunks), oo(n_chunks);
    std::vector<u32> nn(n_chunks), rows(n_chunks);
    u64 soff = 0, toff = 0;
    u32 max_n = 0, max_rows = 0;
    const u64 row_bytes = (u64)(nc_full ? nc_full : nc) * sz;
    for (int i = 0; i < n_chunks; i++) {
        const u64 n = (u64)n_rows[i] * row_bytes;
        if (n >= (1ull << 31)) { set_error("chunk %d is %llu bytes; chunks must be < 2 GiB", i, (unsigned long long)n); return MTS_E_ARG; }
        ic[i].c_off = (u64)c_off[i]; ic[i].c_len = (u64)c_len[i];
        ic[i].stream_off = soff; ic[i].tok_off = toff; ic[i].n_expect = (u32)n;
        ic[i].n_need = nc_full ? (u32)((u64)n_rows[i] * nc * sz) : 0u;
        if (nc_full && ic[i].n_need == 0) ic[i].n_need = 1;      // (a chunk without rows: still a partial decode)
        so[i] = soff; oo[i] = (u64)out_off[i]; nn[i] = (u32)n; rows[i] = (u32)n_rows[i];
        if (n > max_n) max_n = (u32)n;
        if (n_rows[i] > (long)max_rows) max_rows = (u32)n_rows[i];
        soff += align_up(n + STREAM_PAD, STREAM_ALIGN);
        toff += align_up(n + 2, 4);          // 16-byte aligned token arrays (vector loads)
    }
    int rc;
    if ((rc = E.stream.ensure(soff + STREAM_PAD))) return rc;
    if ((rc = E.tokens.ensure((toff + 64) * 4))) return rc;
    // (what the host fills lies in a row: one copy)
    const size_t o_ic = 0, o_so = align_up(sizeof(InfChunk) * n_chunks, 256), o_nn = o_so + align_up(8 * (size_t)n_chunks, 256),
                 o_oo = o_nn + align_up(4 * (size_t)n_chunks, 256), o_rows = o_oo + align_up(8 * (size_t)n_chunks, 256),
                 o_res = o_rows + align_up(4 * (size_t)n_chunks, 256), o_status = o_res + align_up(sizeof(InfResult) * n_chunks, 256),
                 o_end = o_status + align_up(4 * (size_t)n_chunks, 256);
    if ((rc = E.inf_desc.ensure(o_end + 256))) return rc;
    if ((rc = E.adler.ensure(sizeof(u64) * 2 * n_chunks + 256))) return rc;
    if ((rc = E.segsums.ensure(cumsum_scratch_bytes(n_chunks, max_rows, nc)))) return rc;
    if ((rc = E.init_events())) return rc;
    u8 *dp = E.inf_desc.as<u8>();
    InfChunk *d_ic = (InfChunk *)(dp + o_ic);
    InfResult *d_res = (InfResult *)(dp + o_res);
    u64 *d_so = (u64 *)(dp + o_so);
    u64 *d_oo = (u64 *)(dp + o_oo);
    u32 *d_rows = (u32 *)(dp + o_rows);
    int *d_status = (int *)(dp + o_status);
    {
        std::vector<u64> clens(n_chunks);
        for (int i = 0; i < n_chunks; i++) clens[i] = ic[i].c_len;
        if ((rc = E.inf_scratch.ensure(inflate_scratch_bytes(n_chunks, clens.data(), nn.data())))) return rc;
    }
    {
        std::vector<u8> &hst = E.host_stage[0];
        hst.assign(o_res, 0);
        memcpy(hst.data() + o_ic, ic.data(), sizeof(InfChunk) * n_chunks);
        memcpy(hst.data() + o_so, so.data(), 8 * (size_t)n_chunks);
        memcpy(hst.data() + o_nn, nn.data(), 4 * (size_t)n_chunks);
        memcpy(hst.data() + o_oo, oo.data(), 8 * (size_t)n_chunks);
        memcpy(hst.data() + o_rows, rows.data(), 4 * (size_t)n_chunks);
        MTS_HIP(hipMemcpyAsync(dp, hst.data(), o_res, hipMemcpyHostToDevice, st));
    }
    E.t_begin(st);
    if ((rc = launch_inflate(st, d_cdata, d_ic, ic.data(), n_chunks, E.stream.as<u8>(), E.tokens.as<u32>(), d_res, E.adler.as<u64>(),
                             max_n, d_status, E.inf_scratch.p, &E))) return rc;
    if (d_out) {
        if ((rc = launch_cumsum_transpose(st, E.stream.p, d_out, d_so, d_oo, d_rows, d_status, n_chunks, max_rows, nc, sz, flags,
                                          E.segsums.p))) return rc;
        E.t_mark(st, "cumsum_transpose");
    }
    MTS_HIP(hipMemcpyAsync(status, d_status, 4 * (size_t)n_chunks, hipMemcpyDeviceToHost, st));
    MTS_HIP(hipStreamSynchronize(st));
    if (times != 2) E.t_collect(times == 1);
    if (stream_copy_host && nn[0]) MTS_HIP(hipMemcpy(stream_copy_host, E.stream.as<u8>() + so[0], nn[0], hipMemcpyDeviceToHost));
    // A stream that parses to its end, but to another size than the caller expects: the reference inflates it whole and has its
    // adler32 checked before it looks at the size (zlib.decompress raises at mtscomp.py:618-621, the assert comes at :628).  The
    // same order here: such a chunk is inflated once more, alone, at the size it really has, for its check value only -- a
    // valid stream keeps BADSIZE (the assert), a damaged one becomes CORRUPT (the IOError).  It never happens on a file the Writer
    // made; what it costs does not matter.
    if (size_verdict && !nc_full) {
        std::vector<std::pair<int, u32>> odd;
        for (int i = 0; i < n_chunks; i++)
            if (status[i] == MTS_CHUNK_BADSIZE) {
                InfResult r;
                MTS_HIP(hipMemcpy(&r, d_res + i, sizeof(r), hipMemcpyDeviceToHost));
                odd.push_back({i, r.n_out});
            }
        for (const auto &o : odd) {                                  // (from here on the engine's buffers are the verdict passes')
            const int i = o.first;
            if (o.second >= (1u << 31)) { status[i] = MTS_CHUNK_CORRUPT; continue; }     // beyond what a pass can hold: damage, by all odds
            const long rows1 = (long)o.second, off0 = 0;
            int st1 = MTS_CHUNK_CORRUPT;
            const int rc1 = decompress_batch(E, st, d_cdata, c_off + i, c_len + i, &rows1, 1, 1, 1, 0, nullptr, &off0, &st1, 2, nullptr, 0, false);
            if (rc1 == MTS_E_NOMEM) { status[i] = MTS_CHUNK_CORRUPT; continue; }         // (the same call: a size nobody wrote)
            if (rc1) return rc1;
            if (st1 != MTS_CHUNK_OK) status[i] = MTS_CHUNK_CORRUPT;
        }
    }
    return MTS_OK;
}

void inflate_mark(void *engine, hipStream_t st, const char *name) { ((Engine *)engine)->t_mark(st, name); }
u8 *inflate_host_stage(void *engine, size_t bytes) { auto &v = ((Engine *)engine)->host_stage[1]; v.assign(bytes, 0); return v.data(); }

static int dev_decompress(Engine &E, hipStream_t st, const u8 *d_cdata, const long *c_off, const long *c_len, const long *n_rows,
                          int n_chunks, int nc, int sz, int flags, u8 *d_out, const long *out_off, int *status, int nc_full = 0,
                          bool add_times = false)
{
    if (sz != 1 && sz != 2 && sz != 4 && sz != 8) { set_error("itemsize %d unsupported", sz); return MTS_E_ARG; }
    if ((flags & MTS_FLAG_FLOAT) && sz != 4 && sz != 8) { set_error("float items of %d bytes unsupported", sz); return MTS_E_ARG; }
    if (nc <= 0 || n_chunks < 0) return MTS_E_ARG;
    MTS_HIP(hipSetDevice(E.dev));
    const size_t budget = batch_budget_bytes() * 4;          // inflate needs ~5 bytes of workspace per byte
    const u64 row_bytes = (u64)(nc_full > nc ? nc_full : nc) * sz;
    int i = 0;
    bool first = !add_times;
    while (i < n_chunks) {
        int j = i;
        size_t acc = 0;
        while (j < n_chunks) {
            const size_t n = (size_t)n_rows[j] * row_bytes;
            if (j > i && (acc + n > budget || j - i >= MAX_BATCH_CHUNKS)) break;
            acc += n; j++;
        }
        int rc = decompress_batch(E, st, d_cdata, c_off + i, c_len + i, n_rows + i, j - i, nc, sz, flags, d_out, out_off + i,
                                  status + i, first ? 0 : 1, nullptr, nc_full);
        if (rc) return rc;
        first = false;
        i = j;
    }
    if (n_chunks == 0) E.n_stage_done = 0;
    return MTS_OK;
}

}  // namespace mts

using namespace mts;

// ================================================================================================
// extern "C"
// ================================================================================================
extern "C" {

int mts_version(void) { return 100; }

int mts_device_count(void) { return device_count() > 0 ? device_count() : 0; }

const char *mts_strerror(int code)
{
    switch (code) {
    case MTS_OK: return "ok";
    case MTS_E_ARG: return "bad argument";
    case MTS_E_NODEV: return "no usable gfx950 device";
    case MTS_E_HIP: return "HIP runtime error";
    case MTS_E_NOMEM: return "out of memory";
    case MTS_E_UNSUPPORTED: return "not implemented";
    case MTS_E_INTERNAL: return "internal error";
    case MTS_E_MISS: return "chunk not resident in the device cache";
    default: return "unknown error";
    }
}

const char *mts_last_error(void) { return g_err; }

long mts_compress_bound(long raw_len) { return compress_bound(raw_len); }

// ---- decoded-chunk cache on the device (Reader random access) ------------------------------------
namespace {
struct CacheEntry { u8 *d = nullptr; u64 cap = 0, size = 0, stamp = 0; long rows = 0; int cols = 0; };      // (rows, cols) C order; cols < n_channels: the leading channels only
struct DevCache {
    int device = 0;
    u64 capacity = 0, used = 0, clock = 0;
    std::unordered_map<long, CacheEntry> map;
    std::vector<std::pair<u8 *, u64>> spare;         // buffers of evicted entries, reused for new ones
    void drop(long key)
    {
        auto it = map.find(key);
        if (it == map.end()) return;
        used -= it->second.cap;
        if (spare.size() < 4) spare.push_back({it->second.d, it->second.cap}); else (void)hipFree(it->second.d);
        map.erase(it);
    }
    // room for `need` more bytes: evict least recently used entries that this call does not use (stamp < keep_from)
    void make_room(u64 need, u64 keep_from)
    {
        while (used + need > capacity) {
            long victim = 0; u64 best = ~0ull; bool found = false;
            for (auto &kv : map) if (kv.second.stamp < keep_from && kv.second.stamp < best) { best = kv.second.stamp; victim = kv.first; found = true; }
            if (!found) break;                       // everything left belongs to this call: overshoot rather than fail
            drop(victim);
        }
    }
    int alloc(u64 size, u8 **out, u64 *cap)
    {
        for (size_t k = 0; k < spare.size(); k++)
            if (spare[k].second >= size && spare[k].second <= size + size / 2 + 4096) {
                *out = spare[k].first; *cap = spare[k].second; spare.erase(spare.begin() + k); return MTS_OK;
            }
        while (!spare.empty()) { (void)hipFree(spare.back().first); spare.pop_back(); }
        const u64 want = align_up(size ? size : 1, 4096);
        hipError_t e = hipMalloc((void **)out, want);
        if (e != hipSuccess) { set_error("hipMalloc(%llu) for the chunk cache failed: %s", (unsigned long long)want, hipGetErrorString(e)); return MTS_E_NOMEM; }
        *cap = want;
        return MTS_OK;
    }
    void clear()
    {
        for (auto &kv : map) (void)hipFree(kv.second.d);
        map.clear();
        for (auto &b : spare) (void)hipFree(b.first);
        spare.clear();
        used = 0;
    }
};
std::mutex g_cache_mu;
std::unordered_map<long, DevCache *> g_caches;
long g_cache_next = 1;
DevCache *find_cache(long id, int *device = nullptr)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_caches.find(id);
    if (it == g_caches.end()) return nullptr;
    if (device) *device = it->second->device;                  // (read under the lock: the cache may be freed once it is released)
    return it->second;
}
// mts_cache_destroy unregisters a cache first and frees it under its engine's lock; an entry point that looked the cache
// up before it took that lock asks again once it holds it, and never touches a cache that has gone in between
bool cache_alive(long id, const DevCache *c) { return find_cache(id) == c; }
}  // namespace

}  // extern "C"
namespace mts {
void drop_device_caches()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto &kv : g_caches) if (kv.second->device == dev) kv.second->clear();      // (the caller holds this device's engine lock)
}
}  // namespace mts
extern "C" {

void mts_release(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (Engine *e : g_engines)
        if (e) {
            std::lock_guard<std::mutex> l2(e->mu);
            (void)hipSetDevice(e->dev);
            e->release_all();
            std::lock_guard<std::mutex> l3(g_cache_mu);
            for (auto &kv : g_caches) if (kv.second->device == e->dev) kv.second->clear();
        }
}

int mts_dev_compress_chunks(int device, void *stream, const void *d_raw, int n_channels, int itemsize,
                            const long *chunk_bounds, int n_chunks, int flags, int level, unsigned char *d_out,
                            const long *out_slot_offsets, long *out_sizes)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E->mu);
    return dev_compress(*E, (hipStream_t)stream, d_raw, n_channels, itemsize, chunk_bounds, n_chunks, flags, level, d_out,
                        out_slot_offsets, out_sizes);
}

int mts_compress_chunks(int device, const void *raw, int n_channels, int itemsize, const long *chunk_bounds, int n_chunks,
                        int flags, int level, unsigned char *out, const long *out_slot_offsets, long *out_sizes)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (n_chunks <= 0) return n_chunks == 0 ? MTS_OK : MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    const u64 row_bytes = (u64)n_channels * itemsize;
    const u64 raw_bytes = (u64)(chunk_bounds[n_chunks] - chunk_bounds[0]) * row_bytes;
    std::vector<long> slots(n_chunks);
    u64 total = 0;
    for (int i = 0; i < n_chunks; i++) {
        slots[i] = (long)total;
        total += align_up((u64)compress_bound((long)((u64)(chunk_bounds[i + 1] - chunk_bounds[i]) * row_bytes)), 256);
    }
    if ((rc = E->h_in.ensure(raw_bytes + 256))) return rc;
    if ((rc = E->h_out.ensure(total + 256))) return rc;
    const std::vector<int> pb = pipe_pieces(chunk_bounds, true, n_chunks, row_bytes);
    const int np = (int)pb.size() - 1;
    const int dev = E->dev;
    auto copy_in = [&](int k) -> int {                               // the raw rows of piece k (on the calling or on a helper thread)
        MTS_HIP(hipSetDevice(dev));
        const u64 off = (u64)(chunk_bounds[pb[k]] - chunk_bounds[0]) * row_bytes, len = (u64)(chunk_bounds[pb[k + 1]] - chunk_bounds[pb[k]]) * row_bytes;
        return len ? staged_h2d(*E, E->h_in.as<u8>() + off, (const u8 *)raw + off, len) : MTS_OK;
    };
    auto copy_out = [&](int k) -> int {                              // the streams of piece k, each to its slot in the caller's buffer
        MTS_HIP(hipSetDevice(dev));
        std::vector<CopyItem> segs;
        for (int i = pb[k]; i < pb[k + 1]; i++)
            if (out_sizes[i] > 0) segs.push_back({out + out_slot_offsets[i], E->h_out.as<u8>() + slots[i], (size_t)out_sizes[i]});
        return staged_d2h_multi(*E, segs);
    };
    if ((rc = copy_in(0))) return rc;
    for (int k = 0; k < np; k++) {
        std::future<int> f_in, f_out;
        if (k + 1 < np) f_in = copy_beside(copy_in, k + 1);
        if (k >= 1) f_out = copy_beside(copy_out, k - 1);
        const u64 off = (u64)(chunk_bounds[pb[k]] - chunk_bounds[0]) * row_bytes;
        rc = dev_compress(*E, nullptr, E->h_in.as<u8>() + off, n_channels, itemsize, chunk_bounds + pb[k], pb[k + 1] - pb[k], flags, level,
                          E->h_out.as<u8>(), slots.data() + pb[k], out_sizes + pb[k], k > 0);
        const int rc_in = f_in.valid() ? f_in.get() : MTS_OK, rc_out = f_out.valid() ? f_out.get() : MTS_OK;      // (always joined: they hold references to this frame)
        if (rc || rc_in || rc_out) return rc ? rc : rc_in ? rc_in : rc_out;
    }
    return copy_out(np - 1);
}

int mts_delta_transpose(int device, const void *raw, long n_samples, int n_channels, int itemsize, int flags,
                        void *stream_out)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) return MTS_E_ARG;
    if (n_samples < 0 || n_channels <= 0) return MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    const u64 n = (u64)n_samples * n_channels * itemsize;
    if (n == 0) return MTS_OK;
    if (n >= (1ull << 31)) return MTS_E_ARG;
    if ((rc = E->h_in.ensure(n + 256))) return rc;
    if ((rc = E->stream.ensure(n + 2 * STREAM_PAD))) return rc;
    if ((rc = E->desc.ensure(4096))) return rc;
    if ((rc = E->adler.ensure(4096))) return rc;
    ChunkDesc c; memset(&c, 0, sizeof c);
    c.n = (u32)n; c.n_rows = (u32)n_samples;
    MTS_HIP(hipMemcpy(E->h_in.p, raw, n, hipMemcpyHostToDevice));
    MTS_HIP(hipMemcpy(E->desc.p, &c, sizeof c, hipMemcpyHostToDevice));
    if ((rc = launch_delta_transpose(nullptr, E->h_in.p, E->stream.p, E->desc.as<ChunkDesc>(), 1, (u32)n_samples, n_channels,
                                     itemsize, flags, E->adler.as<u64>()))) return rc;
    MTS_HIP(hipMemcpy(stream_out, E->stream.p, n, hipMemcpyDeviceToHost));
    return MTS_OK;
}

int mts_cumsum_transpose(int device, const void *stream, long n_samples, int n_channels, int itemsize, int flags, void *out)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) return MTS_E_ARG;
    if (n_samples < 0 || n_channels <= 0) return MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    const u64 n = (u64)n_samples * n_channels * itemsize;
    if (n == 0) return MTS_OK;
    if (n >= (1ull << 31)) return MTS_E_ARG;
    if ((rc = E->stream.ensure(n + 2 * STREAM_PAD))) return rc;
    if ((rc = E->h_out.ensure(n + 256))) return rc;
    if ((rc = E->desc.ensure(4096))) return rc;
    if ((rc = E->segsums.ensure(cumsum_scratch_bytes(1, (u32)n_samples, n_channels)))) return rc;
    struct { u64 so, oo; u32 rows; } h = {0, 0, (u32)n_samples};
    u8 *dp = E->desc.as<u8>();
    MTS_HIP(hipMemcpy(E->stream.p, stream, n, hipMemcpyHostToDevice));
    MTS_HIP(hipMemcpy(dp, &h.so, 8, hipMemcpyHostToDevice));
    MTS_HIP(hipMemcpy(dp + 8, &h.oo, 8, hipMemcpyHostToDevice));
    MTS_HIP(hipMemcpy(dp + 16, &h.rows, 4, hipMemcpyHostToDevice));
    if ((rc = launch_cumsum_transpose(nullptr, E->stream.p, E->h_out.p, (u64 *)dp, (u64 *)(dp + 8), (u32 *)(dp + 16), nullptr, 1,
                                      (u32)n_samples, n_channels, itemsize, flags, E->segsums.p))) return rc;
    MTS_HIP(hipMemcpy(out, E->h_out.p, n, hipMemcpyDeviceToHost));
    return MTS_OK;
}

int mts_dev_synth_int16(int device, void *stream, void *d_out, long t0, long t1, int n_channels, long seed)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    return launch_synth_int16((hipStream_t)stream, (int16_t *)d_out, t0, t1, n_channels, seed);
}

// ------------------------------------------------------------------------------------------------
// device memory for callers of the dev_* entry points: a process that keeps its recordings in HBM allocates, copies and waits
// through THIS library -- one HIP runtime per process, the one these kernels are launched with (no second runtime's handles)
// ------------------------------------------------------------------------------------------------
__global__ void k_count_diff(const u8 *__restrict__ a, const u8 *__restrict__ b, u64 n, unsigned long long *__restrict__ out /* count, first */)
{
    const u64 n16 = n / 16, stride = (u64)gridDim.x * blockDim.x;
    u32 cnt = 0;
    u64 first = ~0ull;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 x = ((const uint4 *)a)[i], y = ((const uint4 *)b)[i];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) {
            for (int k = 0; k < 16; k++) if (a[i * 16 + k] != b[i * 16 + k]) { cnt++; if (first == ~0ull) first = i * 16 + k; }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (u32)(n & 15)) { const u64 j = n16 * 16 + threadIdx.x; if (a[j] != b[j]) { cnt++; first = first < j ? first : j; } }
    if (cnt) { atomicAdd(&out[0], (unsigned long long)cnt); atomicMin(&out[1], (unsigned long long)first); }
}

int mts_host_alloc(long nbytes, void **h_ptr)
{
    if (nbytes < 0 || !h_ptr) return MTS_E_ARG;
    if (device_count() <= 0) { set_error("no gfx950 device visible (libmtscomp_hip has no CPU path)"); return MTS_E_NODEV; }
    void *p = nullptr;
    if (hipHostMalloc(&p, nbytes > 0 ? (size_t)nbytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc of %ld bytes failed", nbytes); return MTS_E_NOMEM; }
    *h_ptr = p;
    return MTS_OK;
}

int mts_host_free(void *h_ptr)
{
    if (h_ptr) MTS_HIP(hipHostFree(h_ptr));
    return MTS_OK;
}

int mts_dev_alloc(int device, long nbytes, void **d_ptr)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (nbytes < 0 || !d_ptr) return MTS_E_ARG;
    MTS_HIP(hipSetDevice(E->dev));
    void *p = nullptr;
    if (hipMalloc(&p, nbytes > 0 ? (size_t)nbytes : 1) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %ld bytes failed", nbytes); return MTS_E_NOMEM; }
    *d_ptr = p;
    return MTS_OK;
}

int mts_dev_free(int device, void *d_ptr)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    MTS_HIP(hipSetDevice(E->dev));
    if (d_ptr) MTS_HIP(hipFree(d_ptr));
    return MTS_OK;
}

int mts_dev_copy(int device, void *stream, void *dst, const void *src, long nbytes, int kind)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (nbytes < 0 || kind < 0 || kind > 2) return MTS_E_ARG;
    MTS_HIP(hipSetDevice(E->dev));
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (nbytes) MTS_HIP(hipMemcpyAsync(dst, src, (size_t)nbytes, k, (hipStream_t)stream));
    MTS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return MTS_OK;
}

int mts_dev_sync(int device)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    MTS_HIP(hipSetDevice(E->dev));
    MTS_HIP(hipDeviceSynchronize());
    return MTS_OK;
}

int mts_dev_compare(int device, void *stream, const void *d_a, const void *d_b, long nbytes, long *n_diff, long *first_diff)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (nbytes < 0 || !n_diff) return MTS_E_ARG;
    if (((uintptr_t)d_a | (uintptr_t)d_b) & 15) { set_error("mts_dev_compare: buffers must be 16-byte aligned"); return MTS_E_ARG; }
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    unsigned long long *d_out = nullptr, h[2] = {0, ~0ull};
    MTS_HIP(hipMalloc(&d_out, 16));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(d_out, h, 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nbytes) {
        hipLaunchKernelGGL(k_count_diff, dim3(2048), dim3(256), 0, st, (const u8 *)d_a, (const u8 *)d_b, (u64)nbytes, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_out);
    MTS_HIP(e);
    *n_diff = (long)h[0];
    if (first_diff) *first_diff = h[0] ? (long)h[1] : -1;
    return MTS_OK;
}

int mts_last_stage_times(int device, const char **names, float *ms, int cap)
{
    Engine *E;
    if (get_engine(device, &E)) return 0;
    std::lock_guard<std::mutex> lk(E->mu);
    int n = E->n_stage_done < cap ? E->n_stage_done : cap;
    for (int i = 0; i < n; i++) { names[i] = E->done_name[i]; ms[i] = E->stage_ms[i]; }
    return n;
}


int mts_dev_decompress_chunks(int device, void *stream, const unsigned char *d_cdata, const long *c_offsets,
                              const long *c_lengths, const long *n_rows, int n_chunks, int n_channels, int itemsize, int flags,
                              void *d_out, const long *out_offsets, int *chunk_status)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E->mu);
    return dev_decompress(*E, (hipStream_t)stream, d_cdata, c_offsets, c_lengths, n_rows, n_chunks, n_channels, itemsize, flags,
                          (u8 *)d_out, out_offsets, chunk_status);
}

int mts_decompress_chunks(int device, const unsigned char *cdata, const long *c_offsets, const long *c_lengths,
                          const long *n_rows, int n_chunks, int n_channels, int itemsize, int flags, void *out,
                          const long *out_offsets, int *chunk_status)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (n_chunks <= 0) return n_chunks == 0 ? MTS_OK : MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    const u64 row_bytes = (u64)n_channels * itemsize;
    std::vector<long> coff(n_chunks), ooff(n_chunks);
    u64 ctot = 0, otot = 0;
    for (int i = 0; i < n_chunks; i++) {
        if (c_lengths[i] < 0 || n_rows[i] < 0) return MTS_E_ARG;
        coff[i] = (long)ctot; ctot += align_up((u64)c_lengths[i] + 8, 16);
        ooff[i] = (long)otot; otot += align_up((u64)n_rows[i] * row_bytes, 256);
    }
    // the compressed bytes: when the chunks lie (nearly) back to back in the caller's buffer -- a range read from a .cbin --
    // the whole range crosses the bus in one staged copy and the chunks keep their distances; otherwise chunk by chunk
    long lo = c_offsets[0], hi = c_offsets[0] + c_lengths[0];
    u64 sum = 0;
    for (int i = 0; i < n_chunks; i++) {
        if (c_offsets[i] < lo) lo = c_offsets[i];
        if (c_offsets[i] + c_lengths[i] > hi) hi = c_offsets[i] + c_lengths[i];
        sum += (u64)c_lengths[i];
    }
    const bool one_range = lo >= 0 && (u64)(hi - lo) <= sum + sum / 4 + 4096;
    if (one_range) { ctot = (u64)(hi - lo) + 16; for (int i = 0; i < n_chunks; i++) coff[i] = c_offsets[i] - lo; }
    if ((rc = E->h_in.ensure(ctot + 256))) return rc;
    if ((rc = E->h_out.ensure(otot + 256))) return rc;
    // piece by piece (see pipe_pieces) when the compressed chunks lie in file order in one range: piece k's bytes are then one
    // range of the caller's buffer as well
    bool ascending = true;
    for (int i = 1; i < n_chunks && ascending; i++) ascending = c_offsets[i] >= c_offsets[i - 1] + c_lengths[i - 1];
    std::vector<int> pb = {0, n_chunks};
    if (!one_range || ascending) pb = pipe_pieces(n_rows, false, n_chunks, row_bytes);
    const int np = (int)pb.size() - 1;
    const int dev = E->dev;
    auto copy_in = [&](int k) -> int {
        MTS_HIP(hipSetDevice(dev));
        if (one_range) {
            const long a = k == 0 ? lo : c_offsets[pb[k]], b = k + 1 == np ? hi : c_offsets[pb[k + 1]];
            return b > a ? staged_h2d(*E, E->h_in.as<u8>() + (a - lo), cdata + a, (size_t)(b - a)) : MTS_OK;
        }
        for (int i = pb[k]; i < pb[k + 1]; i++)                       // chunks that lie apart in the caller's memory: one copy each
            if (c_lengths[i]) { const int rc1 = staged_h2d(*E, E->h_in.as<u8>() + coff[i], cdata + c_offsets[i], (size_t)c_lengths[i]); if (rc1) return rc1; }
        return MTS_OK;
    };
    auto copy_out = [&](int k) -> int {
        MTS_HIP(hipSetDevice(dev));
        std::vector<CopyItem> segs;
        for (int i = pb[k]; i < pb[k + 1]; i++)
            if (chunk_status[i] == MTS_CHUNK_OK && n_rows[i])
                segs.push_back({(u8 *)out + out_offsets[i], E->h_out.as<u8>() + ooff[i], (size_t)((u64)n_rows[i] * row_bytes)});
        return staged_d2h_multi(*E, segs);
    };
    if ((rc = copy_in(0))) return rc;
    for (int k = 0; k < np; k++) {
        std::future<int> f_in, f_out;
        if (k + 1 < np) f_in = copy_beside(copy_in, k + 1);
        if (k >= 1) f_out = copy_beside(copy_out, k - 1);
        rc = dev_decompress(*E, nullptr, E->h_in.as<u8>(), coff.data() + pb[k], c_lengths + pb[k], n_rows + pb[k], pb[k + 1] - pb[k], n_channels,
                            itemsize, flags, E->h_out.as<u8>(), ooff.data() + pb[k], chunk_status + pb[k], 0, k > 0);
        const int rc_in = f_in.valid() ? f_in.get() : MTS_OK, rc_out = f_out.valid() ? f_out.get() : MTS_OK;
        if (rc || rc_in || rc_out) return rc ? rc : rc_in ? rc_in : rc_out;
    }
    return copy_out(np - 1);
}

// ---- decoded-chunk cache: entry points (state above mts_release) ----------------------------------

int mts_cache_create(int device, long capacity_bytes, long *cache_id)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (!cache_id || capacity_bytes < 0) return MTS_E_ARG;
    DevCache *c = new DevCache();
    c->device = device; c->capacity = (u64)capacity_bytes;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    *cache_id = g_cache_next++;
    g_caches[*cache_id] = c;
    return MTS_OK;
}

int mts_cache_destroy(long cache_id)
{
    DevCache *c;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_caches.find(cache_id);
        if (it == g_caches.end()) return MTS_E_ARG;
        c = it->second;
        g_caches.erase(it);                                    // from here on no entry point starts on this cache; those inside finish first (engine lock)
    }
    Engine *E;
    if (get_engine(c->device, &E) == MTS_OK) {
        std::lock_guard<std::mutex> lk(E->mu);
        (void)hipSetDevice(E->dev);
        c->clear();
    }
    delete c;
    return MTS_OK;
}

int mts_cache_query(long cache_id, const long *chunk_keys, int n, int *present)
{
    int dev = 0;
    DevCache *c = find_cache(cache_id, &dev);
    if (!c || n < 0) return MTS_E_ARG;
    Engine *E;
    int rc = get_engine(dev, &E);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(E->mu);
    if (!cache_alive(cache_id, c)) return MTS_E_ARG;
    for (int i = 0; i < n; i++) {           // 0: not resident, else the number of (leading) channels the entry holds
        auto it = c->map.find(chunk_keys[i]);
        present[i] = it == c->map.end() ? 0 : it->second.cols > 0 ? it->second.cols : 1;
    }
    return MTS_OK;
}

// make every listed chunk resident (decode the missing ones in one batch) and pin them for this call by their stamp
static int cache_ensure(DevCache *c, Engine *E, int n_chunks, const long *chunk_keys, const unsigned char *cdata, const long *c_offsets,
                        const long *c_lengths, const long *n_rows, int n_channels, int itemsize, int flags, int *chunk_status, u64 call_stamp,
                        long *total_rows_out, int n_cols /* leading channels wanted: n_channels = whole chunks */)
{
    int rc;
    std::vector<int> miss;
    long total_rows = 0;
    {   // every key once: a key listed twice would be decoded and accounted twice
        std::vector<long> keys(chunk_keys, chunk_keys + n_chunks);
        std::sort(keys.begin(), keys.end());
        if (std::adjacent_find(keys.begin(), keys.end()) != keys.end()) { set_error("a chunk key is listed twice"); return MTS_E_ARG; }
    }
    auto usable = [&](int i) -> bool {
        auto it = c->map.find(chunk_keys[i]);
        return it != c->map.end() && it->second.rows == n_rows[i] && it->second.cols >= n_cols && it->second.cols <= n_channels &&
               it->second.size == (u64)n_rows[i] * it->second.cols * itemsize;
    };
    for (int i = 0; i < n_chunks; i++) {          // every key is looked at before anything is dropped: a miss leaves the cache as it was
        if (n_rows[i] < 0) return MTS_E_ARG;
        if (!usable(i) && c_lengths[i] <= 0) {
            set_error("chunk key %ld is not resident%s and no compressed bytes were given", chunk_keys[i], c->map.count(chunk_keys[i]) ? " with the channels asked for" : "");
            return MTS_E_MISS;
        }
    }
    for (int i = 0; i < n_chunks; i++) {
        total_rows += n_rows[i];
        chunk_status[i] = MTS_CHUNK_OK;
        if (usable(i)) { c->map.find(chunk_keys[i])->second.stamp = call_stamp; continue; }
        c->drop(chunk_keys[i]);                    // same key, other shape or fewer channels: decoded again
        miss.push_back(i);
    }
    *total_rows_out = total_rows;
    auto all_resident = [&]() -> int {            // (a workspace allocation that failed may have emptied the caches of this device)
        for (int i = 0; i < n_chunks; i++)
            if (chunk_status[i] == MTS_CHUNK_OK && !c->map.count(chunk_keys[i])) { set_error("chunk key %ld was dropped from the cache during the call", chunk_keys[i]); return MTS_E_MISS; }
        return MTS_OK;
    };
    if (miss.empty()) return all_resident();
    static const bool times = getenv("MTS_CACHE_TIMES") != nullptr;      // (where a cold read's time goes: stderr, one line per call)
    const auto t_0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count(); };
    double t_h2d = 0, t_dec = 0;
    const int m = (int)miss.size();
    const u64 row_bytes = (u64)n_cols * itemsize;              // of what is decoded and kept
    std::vector<long> coff(m), clen(m), rows(m), ooff(m);
    std::vector<int> st(m);
    u64 ctot = 0, otot = 0;
    for (int k = 0; k < m; k++) {
        const int i = miss[k];
        clen[k] = c_lengths[i]; rows[k] = n_rows[i];
        // (chunks that lie back to back in the caller's buffer keep their distances on the device: one copy moves the run)
        const bool joins = k > 0 && c_offsets[i] == c_offsets[miss[k - 1]] + clen[k - 1];
        if (!joins) ctot = align_up(ctot + (k ? 16 : 0), 16);
        coff[k] = (long)ctot; ctot += (u64)clen[k];
        ooff[k] = (long)otot; otot += align_up((u64)rows[k] * row_bytes, 256);
    }
    ctot += 16;
    if ((rc = E->h_in.ensure(ctot + 256))) return rc;
    if ((rc = E->h_out.ensure(otot + 256))) return rc;
    // the compressed bytes: chunks that lie back to back in the caller's buffer (a range read or mapped from a .cbin) cross in ONE
    // staged copy -- page-locked memory by DMA as it is, anything else (a mapping of the file, a bytes object) through the
    // page-locked pieces, copied by the host threads while the DMA of the piece before runs
    for (int k = 0; k < m;) {
        int j = k + 1;
        while (j < m && c_offsets[miss[j]] == c_offsets[miss[j - 1]] + clen[j - 1] && coff[j] == coff[j - 1] + clen[j - 1]) j++;
        u64 len = 0;
        for (int q = k; q < j; q++) len += (u64)clen[q];
        if (len && (rc = staged_h2d(*E, E->h_in.as<u8>() + coff[k], cdata + c_offsets[miss[k]], (size_t)len))) return rc;
        k = j;
    }
    if (times) t_h2d = since();
    rc = dev_decompress(*E, nullptr, E->h_in.as<u8>(), coff.data(), clen.data(), rows.data(), m, n_cols, itemsize, flags,
                        E->h_out.as<u8>(), ooff.data(), st.data(), n_channels);
    if (rc) return rc;
    if (times) t_dec = since();
    for (int k = 0; k < m; k++) {
        const int i = miss[k];
        if (st[k] == MTS_CHUNK_NEEDMORE) {
            set_error("chunk key %ld: the %ld compressed bytes given do not reach the %d leading channels asked for", chunk_keys[i], clen[k], n_cols);
            return MTS_E_MISS;
        }
        chunk_status[i] = st[k];
        if (st[k] != MTS_CHUNK_OK) continue;
        const u64 size = (u64)rows[k] * row_bytes;
        CacheEntry e;
        c->make_room(align_up(size ? size : 1, 4096), call_stamp);
        if ((rc = c->alloc(size, &e.d, &e.cap))) return rc;
        e.size = size; e.rows = rows[k]; e.cols = n_cols; e.stamp = call_stamp;
        if (size) MTS_HIP(hipMemcpyAsync(e.d, E->h_out.as<u8>() + ooff[k], (size_t)size, hipMemcpyDeviceToDevice, nullptr));
        c->used += e.cap;
        c->map[chunk_keys[i]] = e;
    }
    if (times) { (void)hipStreamSynchronize(nullptr); fprintf(stderr, "[cache] %d chunks: copy in %.3f ms, decode %.3f ms, entries %.3f ms\n", m, t_h2d, t_dec - t_h2d, since() - t_dec); }
    return all_resident();
}

int mts_cache_read_rows(long cache_id, int n_chunks, const long *chunk_keys, const unsigned char *cdata, const long *c_offsets,
                        const long *c_lengths, const long *n_rows, int n_channels, int itemsize, int flags, long row_begin,
                        long row_end, void *out, int *chunk_status)
{
    int dev = 0;
    DevCache *c = find_cache(cache_id, &dev);
    if (!c || n_chunks < 0 || n_channels <= 0 || row_begin < 0 || row_end < row_begin) return MTS_E_ARG;
    Engine *E;
    int rc = get_engine(dev, &E);
    if (rc) return rc;
    if (n_chunks == 0) return row_end == 0 ? MTS_OK : MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    if (!cache_alive(cache_id, c)) return MTS_E_ARG;         // (destroyed while this call waited for the engine)
    MTS_HIP(hipSetDevice(E->dev));
    const u64 row_bytes = (u64)n_channels * itemsize;
    const u64 call_stamp = ++c->clock;
    long total_rows = 0;
    if ((rc = cache_ensure(c, E, n_chunks, chunk_keys, cdata, c_offsets, c_lengths, n_rows, n_channels, itemsize, flags, chunk_status, call_stamp, &total_rows, n_channels))) return rc;
    if (row_end > total_rows) return MTS_E_ARG;
    // rows [row_begin, row_end) of the concatenation, straight from the resident chunks
    long r0 = 0;
    for (int i = 0; i < n_chunks; i++) {
        const long r1 = r0 + n_rows[i];
        const long lo = row_begin > r0 ? row_begin : r0, hi = row_end < r1 ? row_end : r1;
        if (lo < hi && chunk_status[i] == MTS_CHUNK_OK) {
            const CacheEntry &e = c->map[chunk_keys[i]];
            MTS_HIP(hipMemcpyAsync((u8 *)out + (u64)(lo - row_begin) * row_bytes, e.d + (u64)(lo - r0) * row_bytes,
                                   (size_t)((u64)(hi - lo) * row_bytes), hipMemcpyDeviceToHost, nullptr));
        }
        r0 = r1;
    }
    MTS_HIP(hipStreamSynchronize(nullptr));
    c->make_room(0, ~0ull);                         // back under the capacity (this call's chunks may go too)
    return MTS_OK;
}

int mts_cache_read_slices_leading(long cache_id, int n_chunks, const long *chunk_keys, const unsigned char *cdata, const long *c_offsets,
                                  const long *c_lengths, const long *n_rows, int n_channels, int itemsize, int flags, int n_leading,
                                  int n_req, const long *req, void *out, const long *out_offsets, long out_bytes, int *chunk_status)
{
    if (n_leading <= 0 || n_leading > n_channels) return MTS_E_ARG;
    if (n_leading < n_channels && (!(flags & MTS_FLAG_ORDER_F) || (flags & MTS_FLAG_FLOAT))) {
        set_error("leading channels alone can only be decoded from channel-major integer chunks");
        return MTS_E_ARG;
    }
    int dev = 0;
    DevCache *c = find_cache(cache_id, &dev);
    if (!c || n_chunks < 0 || n_channels <= 0 || n_req < 0 || out_bytes < 0) return MTS_E_ARG;
    if (itemsize != 1 && itemsize != 2 && itemsize != 4 && itemsize != 8) return MTS_E_ARG;
    Engine *E;
    int rc = get_engine(dev, &E);
    if (rc) return rc;
    if (n_chunks == 0 || n_req == 0) return MTS_OK;
    std::lock_guard<std::mutex> lk(E->mu);
    if (!cache_alive(cache_id, c)) return MTS_E_ARG;
    MTS_HIP(hipSetDevice(E->dev));
    const u64 call_stamp = ++c->clock;
    // the requests first: their sizes are known without the chunks, and every allocation of this call has to come BEFORE the
    // residency check -- a workspace allocation that fails once drops this device's decoded chunks (DBuf::ensure)
    long total_rows = 0;
    for (int i = 0; i < n_chunks; i++) { if (n_rows[i] < 0) return MTS_E_ARG; total_rows += n_rows[i]; }
    std::vector<GatherReq> gr(n_req);
    u64 max_items = 0;
    for (int k = 0; k < n_req; k++) {
        const long *q = req + 6 * k;
        if (q[0] < 0 || q[1] < q[0] || q[1] > total_rows || q[2] < 1 || q[3] < 0 || q[4] < q[3] || q[4] > n_leading || q[5] < 1) return MTS_E_ARG;
        GatherReq &g = gr[k];
        g.rb = q[0]; g.rs = q[2]; g.cb = q[3]; g.cs = q[5];
        g.nr = (q[1] - q[0] + q[2] - 1) / q[2]; g.ncol = (q[4] - q[3] + q[5] - 1) / q[5];
        g.out_off = out_offsets[k];
        if (out_offsets[k] < 0 || (u64)out_offsets[k] + (u64)g.nr * g.ncol * itemsize > (u64)out_bytes) return MTS_E_ARG;
        if (out_offsets[k] % itemsize) { set_error("request %d: output offset %ld is not a multiple of the item size", k, out_offsets[k]); return MTS_E_ARG; }
        if ((u64)g.nr * g.ncol > max_items) max_items = (u64)g.nr * g.ncol;
    }
    const size_t o_req = align_up(sizeof(GatherChunk) * n_chunks, 256), desc = o_req + align_up(sizeof(GatherReq) * n_req, 256);
    if ((rc = E->misc.ensure(desc + 256))) return rc;
    if ((rc = E->h_out.ensure((u64)out_bytes + 256))) return rc;
    long total_rows_seen = 0;
    if ((rc = cache_ensure(c, E, n_chunks, chunk_keys, cdata, c_offsets, c_lengths, n_rows, n_channels, itemsize, flags, chunk_status, call_stamp, &total_rows_seen, n_leading))) return rc;
    // (cache_ensure ends with the residency check and nothing below allocates: the base pointers stay valid)
    std::vector<GatherChunk> gc(n_chunks);
    long r0 = 0;
    for (int i = 0; i < n_chunks; i++) {
        gc[i].row0 = r0; r0 += n_rows[i];
        if (chunk_status[i] == MTS_CHUNK_OK) { const CacheEntry &e = c->map[chunk_keys[i]]; gc[i].base = e.d; gc[i].pitch = e.cols; }
        else { gc[i].base = nullptr; gc[i].pitch = n_channels; }
    }
    MTS_HIP(hipMemcpyAsync(E->misc.p, gc.data(), sizeof(GatherChunk) * n_chunks, hipMemcpyHostToDevice, nullptr));
    MTS_HIP(hipMemcpyAsync(E->misc.as<u8>() + o_req, gr.data(), sizeof(GatherReq) * n_req, hipMemcpyHostToDevice, nullptr));
    if (max_items && (rc = launch_gather_slices(nullptr, (const GatherChunk *)E->misc.p, n_chunks, (const GatherReq *)(E->misc.as<u8>() + o_req), n_req, max_items,
                                                n_channels, itemsize, E->h_out.as<u8>()))) return rc;
    if (out_bytes) MTS_HIP(hipMemcpyAsync(out, E->h_out.p, (size_t)out_bytes, hipMemcpyDeviceToHost, nullptr));      // the requested items, nothing else, in one copy
    MTS_HIP(hipStreamSynchronize(nullptr));
    c->make_room(0, ~0ull);
    return MTS_OK;
}

int mts_cache_read_slices(long cache_id, int n_chunks, const long *chunk_keys, const unsigned char *cdata, const long *c_offsets,
                          const long *c_lengths, const long *n_rows, int n_channels, int itemsize, int flags, int n_req,
                          const long *req, void *out, const long *out_offsets, long out_bytes, int *chunk_status)
{
    return mts_cache_read_slices_leading(cache_id, n_chunks, chunk_keys, cdata, c_offsets, c_lengths, n_rows, n_channels, itemsize, flags, n_channels,
                                         n_req, req, out, out_offsets, out_bytes, chunk_status);
}

int mts_debug_inflate(int device, const unsigned char *zbytes, long zlen, unsigned char *out, long out_cap, long *out_len, int *status)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (zlen < 0 || out_cap < 0) return MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    if ((rc = E->h_in.ensure((u64)zlen + 256))) return rc;
    if (zlen) MTS_HIP(hipMemcpy(E->h_in.p, zbytes, (size_t)zlen, hipMemcpyHostToDevice));
    // the expected size is the caller's out_cap: status BADSIZE when the stream inflates to anything else
    const long coff = 0, clen = zlen, rows = out_cap, ooff = 0;
    rc = decompress_batch(*E, nullptr, E->h_in.as<u8>(), &coff, &clen, &rows, 1, 1, 1, 0, nullptr, &ooff, status, 0, out);
    if (rc) return rc;
    if (out_len) *out_len = *status == MTS_CHUNK_OK ? out_cap : 0;
    return MTS_OK;
}

// ---- debug taps ---------------------------------------------------------------------------------
static int debug_compress_stream(int device, const void *stream_bytes, long n, int level, unsigned char *out, long out_cap,
                                 long *out_len, DebugTap *tap)
{
    Engine *E;
    int rc = get_engine(device, &E);
    if (rc) return rc;
    if (level == -1) level = 6;
    if (level < 1 || level > 9) return MTS_E_ARG;
    if (level < 4 && tap && tap->t_full) return MTS_E_UNSUPPORTED;      // (deflate_fast has no candidate tables)
    if (n < 0 || n >= (1l << 31)) return MTS_E_ARG;
    std::lock_guard<std::mutex> lk(E->mu);
    MTS_HIP(hipSetDevice(E->dev));
    const u64 bound = align_up((u64)compress_bound(n), 256);
    if ((rc = E->h_in.ensure((u64)n + 2 * STREAM_PAD))) return rc;
    if ((rc = E->h_out.ensure(bound + 256))) return rc;
    MTS_HIP(hipMemset(E->h_in.p, 0, (u64)n + 2 * STREAM_PAD));
    if (n) MTS_HIP(hipMemcpy(E->h_in.p, stream_bytes, (size_t)n, hipMemcpyHostToDevice));
    const long bounds[2] = {0, n};
    const long slot = 0;
    long size = 0;
    rc = compress_batch(*E, nullptr, E->h_in.as<u8>(), true, 1, 1, bounds, 1, 0, level, E->h_out.as<u8>(), &slot, &size, false, tap);
    if (rc) return rc;
    if (out_len) *out_len = size;
    if (out) {
        if (size > out_cap) return MTS_E_ARG;
        MTS_HIP(hipMemcpy(out, E->h_out.p, (size_t)size, hipMemcpyDeviceToHost));
    }
    return MTS_OK;
}

int mts_debug_match_tables(int device, const void *stream_bytes, long n, int level, unsigned *t_full, unsigned *t_quarter)
{
    DebugTap tap; tap.t_full = t_full; tap.t_quarter = t_quarter;
    return debug_compress_stream(device, stream_bytes, n, level, nullptr, 0, nullptr, &tap);
}
int mts_debug_tokens(int device, const void *stream_bytes, long n, int level, unsigned short *tokens, long *n_tokens)
{
    DebugTap tap; tap.tokens = tokens; tap.n_tokens = n_tokens;
    return debug_compress_stream(device, stream_bytes, n, level, nullptr, 0, nullptr, &tap);
}
int mts_debug_deflate(int device, const void *stream_bytes, long n, int level, unsigned char *out, long out_cap, long *out_len)
{
    return debug_compress_stream(device, stream_bytes, n, level, out, out_cap, out_len, nullptr);
}

}  // extern "C"
