// Internal declarations shared by the HIP translation units of libmtscomp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mtscomp_hip.h"

namespace mts {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ---- DEFLATE constants (zlib 1.2.11, windowBits 15, memLevel 8) --------------------------------
constexpr int MIN_MATCH = 3;
constexpr int MAX_MATCH = 258;
constexpr int WSIZE = 32768;
constexpr int MAX_DIST = WSIZE - 262;      // 32506
constexpr int TOO_FAR = 4096;
constexpr int BLOCK_TOKENS = 16383;        // lit_bufsize - 1
constexpr int L_CODES = 286;
constexpr int D_CODES = 30;
constexpr int BL_CODES = 19;

// ---- stream layout ------------------------------------------------------------------------------
// Every chunk's transformed byte stream lives in one device buffer at a 256-B aligned offset and is
// followed by >= STREAM_PAD zero bytes, so kernels may over-read past a stream's end.
constexpr int STREAM_PAD = 512;
constexpr int STREAM_ALIGN = 256;

// ---- match stage tiling ---------------------------------------------------------------------------
// A tile's history (HALO) is sorted and staged again by the next tile, so bigger tiles mean less work.  While every
// workgroup read its own window through L2, big tiles lost to locality (round 1: 64..96 Ki 41-42 ms, 160 Ki 53, 224 Ki 60);
// with the workgroups of an XCD sharing a tile the window is L2 resident whatever its size and the biggest tile the 18-bit
// window-relative positions allow wins (match 31 -> 28.5 ms, sort 12.9 -> 11.4 from 96 Ki to 224 Ki).
// The 32-bit sort keys hold the window-relative position (REL_BITS allow windows up to 2^18) and the 7 hash bits
// the second radix pass still needs; the first pass takes its 8 bits straight from the bytes.
#ifndef MTS_REL_BITS
#define MTS_REL_BITS 18
#endif
constexpr int REL_BITS = MTS_REL_BITS;     // 18 or 19 (a match entry word holds rel : 9 bits : the low bits of byte 7)
constexpr int HALO = 32768;                // history a tile additionally needs (>= MAX_DIST)
constexpr int WIN = 1 << REL_BITS;         // hashed window of a tile
#ifndef MTS_TILE
#define MTS_TILE (WIN - HALO)
#endif
constexpr int TILE = MTS_TILE;             // positions a match-stage workgroup owns (229376 / 491520; anything up to WIN - HALO)
static_assert(TILE > 0 && TILE + HALO <= WIN && TILE % 64 == 0, "a tile and its history fit the window");
constexpr u32 REL_MASK = (1u << REL_BITS) - 1;

constexpr int SEG = 1024;                  // parse segment (positions per speculative walker); measured 512: 9.0, 1024: 8.5, 2048: 8.9, 4096: 9.3 ms (fixpoint + emit)

struct LevelCfg { int good, lazy, nice, chain; };

// one match-stage tile
struct TileDesc {
    u64 stream_off;      // byte offset of the chunk's stream in the stream buffer
    u64 sorted_off;      // entry offset of this tile's sorted window in the sort buffers
    u32 n;               // stream length (bytes) of the chunk
    u32 a;               // first owned position
    u32 w;               // window start (= max(0, a - HALO))
    u32 wlen;            // hashed positions in the window: positions [w, w + wlen), all <= n - 3
    u32 own_end;         // owned positions are [a, own_end)
    u32 chunk;
};

// per-chunk descriptor of a compress batch
struct ChunkDesc {
    u64 stream_off;      // into the stream buffer
    u64 tok_off;         // into the token buffer (capacity n + 1 tokens)
    u64 out_off;         // byte offset of the chunk's slot in the output buffer (16-B aligned)
    u64 raw_off;         // byte offset of the chunk's first row in the raw input
    u32 n;               // stream bytes
    u32 n_rows;
    u32 seg0;            // first parse segment (global index)
    u32 nseg;
    u32 blk0;            // first block slot (global index); capacity n / 16383 + 2
    u32 blk_cap;
    u32 tile0;           // first match-stage tile of the chunk (global index)
    u32 pad;
};

// per-block record produced by the tree stage
struct BlockRec {
    u32 tok0, ntok;          // token range (chunk-relative)
    u32 in_start, in_len;    // input byte range
    u32 nbits;               // bits of the block incl. the 3 header bits (stored: payload handled apart)
    u32 btype;               // 0 stored 1 fixed 2 dynamic
    u32 hdr_bits;            // dynamic: bits of the tree header (after the 3-bit block header)
    u32 last;
    u64 bit_start;           // bit offset in the chunk's zlib stream
};

struct ChunkOut {
    u64 nbytes;              // compressed size
    u32 ntok, nblk;
    u32 adler;
    u32 trailing;            // last token is the post-loop literal
};

inline __host__ __device__ u64 align_up(u64 x, u64 a) { return (x + a - 1) / a * a; }

// wave-wide predicates straight from the condition's lane mask (HIP's __ballot / __any take an int: the mask is first turned into a
// value per lane -- v_cndmask -- and compared again -- v_cmp: two vector instructions per use in kernels that are bound by them)
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// ---- error plumbing -------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define MTS_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            mts::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                       \
            return MTS_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the CURRENT DEVICE's function object: it is applied once per
// (device, kernel), under a lock (several host threads drive several devices of one process; api.hip)
int ensure_dynamic_lds(const void *kernel, int bytes);
#define MTS_LDS_ATTR(kernel, bytes)                                                 \
    do {                                                                            \
        int rc_ = mts::ensure_dynamic_lds((const void *)(kernel), (int)(bytes));    \
        if (rc_) return rc_;                                                        \
    } while (0)

// ---- kernel launchers (one per stage; all asynchronous on `st`) -----------------------------------
// transform.hip
int launch_delta_transpose(hipStream_t st, const void *d_raw, void *d_stream, const ChunkDesc *d_chunks,
                           int n_chunks, u32 max_rows, int n_channels, int itemsize, int flags,
                           u64 *d_adler_acc /* 2 per chunk, zeroed by the launcher */);
int launch_cumsum_transpose(hipStream_t st, const void *d_stream, void *d_out, const u64 *d_stream_off,
                            const u64 *d_out_off, const u32 *d_rows, const int *d_status, int n_chunks,
                            u32 max_rows, int n_channels, int itemsize, int flags, void *d_segsums);
size_t cumsum_scratch_bytes(int n_chunks, u32 max_rows, int n_channels);
int launch_synth_int16(hipStream_t st, int16_t *d_out, long t0, long t1, int n_channels, long seed);
int launch_adler_stream(hipStream_t st, const u8 *d_stream, const u64 *d_stream_off, const u32 *d_n,
                        int n_chunks, u32 max_n, u64 *d_adler_acc,
                        const u32 *d_skip /* null, or per chunk (stride in words): >= 2 = already summed; the sums are then NOT zeroed here */, u32 skip_stride);

// pieces of decoded chunks gathered on the device (mts_cache_read_slices)
struct GatherChunk { long row0; const u8 *base; long pitch; };          // first row in the concatenation; null = the chunk failed; items per row of the entry
struct GatherReq { long rb, rs, cb, cs, nr, ncol, out_off; };           // rows rb + i * rs (i < nr), columns cb + j * cs (j < ncol) -> out_off
int launch_gather_slices(hipStream_t st, const GatherChunk *d_chunks, int n_chunks, const GatherReq *d_req, int n_req, u64 max_items,
                         int n_channels, int itemsize, u8 *d_out);

// deflate.hip
size_t hash_sort_ws_bytes(int n_tiles);                            // the one-pass sort's per-tile records
int launch_hash_sort(hipStream_t st, const u8 *d_stream, const TileDesc *d_tiles, int n_tiles, u32 *d_tmp, u32 *d_sorted,
                     int force_ballot, void *d_ws);
constexpr int MATCH_SINK_BYTES = 65536;         // behind the flag words, 256 bytes after the first: where the lanes of the match stage that own no position store (match.hip)
int launch_match(hipStream_t st, const u8 *d_stream, const TileDesc *d_tiles, int n_tiles, const u32 *d_sorted, u32 *d_tables, u32 *d_quarter,
                 LevelCfg cfg, u32 *d_flags /* [0] |= 1: a hash run out of position order; [63 ...]: MATCH_SINK_BYTES of sink */,
                 int all_quarters /* debug tap: write the side table for every position */);
struct ParseBufs {
    u32 *entry, *exit_a, *exit_b, *cnt, *tokbase;   // per segment
    u32 *cp;                                        // per segment 16 words: 7 checkpoint positions, 7 token counts
    u32 *marks;                                     // per segment MARK_WORDS words: what the walk did at every position (deflate.hip: MarkW)
    u32 *seg_chunk, *seg_start;                     // per segment: owning chunk / start position
    int *changed;                                   // device flag
};
// d_tables: one word per stream position (deflate.hip: te_pack), d_quarter: the side table of the quarter-budget results
int launch_parse_spec(hipStream_t st, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks, ParseBufs pb,
                      int n_segs, LevelCfg cfg, int n_chunks, u32 max_nseg /* segments of the longest chunk */);
int launch_parse_fix(hipStream_t st, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks, ParseBufs pb,
                     int n_segs, LevelCfg cfg, int round);
int launch_parse_fix_serial(hipStream_t st, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks, ParseBufs pb, int n_chunks,
                            LevelCfg cfg, int rounds_done);
int launch_parse_count(hipStream_t st, const u32 *d_tables, const ChunkDesc *d_chunks, ParseBufs pb,
                       int n_segs, int n_chunks, LevelCfg cfg, ChunkOut *d_cout);
int launch_parse_emit_marks(hipStream_t st, const u8 *d_stream, const u32 *d_tables, const u32 *d_quarter, const ChunkDesc *d_chunks,
                            ParseBufs pb, int rounds_done, u32 *d_tokens, u32 *d_blk_in_start, ChunkOut *d_cout, int n_chunks, u32 max_nseg);
size_t parse_marks_words(size_t n_segs);
size_t parse_cp_words();                                          // words of checkpoints per segment
int launch_block_trees(hipStream_t st, const ChunkDesc *d_chunks, const u32 *d_blk_chunk, int total_blk_cap,
                       const u32 *d_tokens, const u32 *d_blk_in_start, const ChunkOut *d_cout,
                       BlockRec *d_blocks, u32 *d_blk_codes, u32 *d_blk_hdr, int fast /* levels 1..3: deflate_fast's flush points */);
// levels 1..3 (deflate_fast): inverse map of the sorted order, the rounds of the speculative greedy walk, its in-order completion, the token pass
int launch_inverse_map(hipStream_t st, const u8 *d_stream, const TileDesc *d_tiles, int n_tiles, const u32 *d_sorted, u32 *d_inv, u32 *d_flags);
size_t fast_seq_state_bytes(int n_chunks);
int fast_list_len(int level);           // members per position in the candidate lists of levels 1..3
int fast_list_rows(int level);          // words per position (members + masks + header)
int launch_fast_cands(hipStream_t st, const u8 *d_stream, const ChunkDesc *d_chunks, const TileDesc *d_tiles, const u32 *d_sorted, const u32 *d_inv,
                      u32 *d_lists, u32 W, u32 phase, int n_chunks, int level, LevelCfg cfg);
int launch_fast_seq(hipStream_t st, const u8 *d_stream, const ChunkDesc *d_chunks, const TileDesc *d_tiles, const u32 *d_sorted, const u32 *d_inv,
                    const u32 *d_lists, u32 W, u32 phase, void *d_state, int n_chunks, int level, LevelCfg cfg, u32 *d_tokens, u32 *d_blk_in_start,
                    ChunkOut *d_cout);
int launch_block_layout(hipStream_t st, const ChunkDesc *d_chunks, int n_chunks, BlockRec *d_blocks,
                        ChunkOut *d_cout, const u64 *d_adler_acc);
int launch_block_pack(hipStream_t st, const u8 *d_stream, const ChunkDesc *d_chunks, const u32 *d_blk_chunk,
                      int total_blk_cap, const u32 *d_tokens, const BlockRec *d_blocks,
                      const u32 *d_blk_codes, const u32 *d_blk_hdr, const ChunkOut *d_cout, u8 *d_out,
                      int level);
int launch_zero_edges(hipStream_t st, const ChunkDesc *d_chunks, const u32 *d_blk_chunk, int total_blk_cap, const BlockRec *d_blocks,
                      const ChunkOut *d_cout, u8 *d_out);      // the words of the output the packer ORs into
constexpr int BLK_CODE_WORDS = 320;     // per block: 286 lit/len + 30 dist (code | len << 16), padded
constexpr int BLK_HDR_WORDS = 96;       // per block: packed dynamic-tree header bits (<= 3072 bits)

// inflate.hip
struct InfChunk {
    u64 c_off;           // compressed bytes offset in d_cdata
    u64 c_len;
    u64 stream_off;      // where the inflated stream goes (stream buffer)
    u64 tok_off;         // token buffer offset (capacity n + 2)
    u32 n_expect;        // expected inflated size (the whole chunk)
    u32 n_need;          // 0, or: only the first n_need bytes of the stream are wanted (the leading channels of a channel-major
                         // chunk, for Reader[rows, columns]): the block chain stops once it has them, c_len may be a prefix of
                         // the chunk's bytes, no adler32 check; MTS_CHUNK_NEEDMORE when the bytes given do not get that far
};
constexpr int MTS_CHUNK_NEEDMORE = 1;      // internal per-chunk status (never leaves the library: the cache answers MTS_E_MISS)
struct InfResult {
    int status;          // MTS_CHUNK_*
    u32 n_out;
    u32 ntok;
    u32 adler_stored;
    u64 end_bit;
};
int launch_inflate(hipStream_t st, const u8 *d_cdata, const InfChunk *d_chunks, const InfChunk *h_chunks, int n_chunks,
                   u8 *d_stream, u32 *d_tokens, InfResult *d_res, u64 *d_adler_acc, u32 max_n, int *d_status_out,
                   void *d_scratch, void *engine);
size_t inflate_scratch_bytes(int n_chunks, const u64 *c_lens, const u32 *n_expect);
void inflate_mark(void *engine, hipStream_t st, const char *name);   // stage timing hook (api.hip)
u8 *inflate_host_stage(void *engine, size_t bytes);                    // zeroed host bytes the engine keeps until the next batch (api.hip)

}  // namespace mts
