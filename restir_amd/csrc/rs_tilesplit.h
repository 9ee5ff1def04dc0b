// rs_tilesplit.h -- heavy tiles of the closest-hit kernels as four waves instead of one.
//
// A packet walk (rs_scene.h trace_closest_packet) is a chain of dependent node fetches as long as the UNION of the nodes its 64
// rays visit, and a launch lasts at least as long as its longest chain: on the Bistro-class scene the mean tile has 270 union
// nodes and the worst 3 669, which alone is the 1.4 ms of its primary-ray kernel; a 1/8 strip of any scene is a single round of
// waves and ends with its slowest tile.  tools/walk_stats.py: in tiles of >= 1024 union nodes the slowest RAY visits 598 nodes
// against 1 503 for the wave -- those tiles are heavy because their rays diverge, so a quarter of the rays is a much shorter chain.
//
// Each lane's own walk does not depend on which other rays share its wave, so regrouping rays changes no result.  What is used to
// regroup them is how long the tile's chain was the LAST time the same launch ran on this stream (a scheduling hint, not a result:
// every pixel is traced again in full):
//   * every wave reports its union-node count; a regular wave whose count reaches `threshold` appends its tile to the out-list and
//     sets the tile's flag;
//   * the next launch of the same geometry on that stream reads them: `helperBlocks` blocks at the FRONT of the grid (so that the
//     long chains start first) take one listed tile each, its four waves a 4x4 quadrant (16 lanes) each, and the regular wave of a
//     flagged tile does nothing.  The helper block keeps the tile listed while the sum of its quadrant counts stays at the threshold.
// A tile is flagged iff it is in the first `capacity` list entries, both are written by the same kernel and read by a later launch
// on the same stream, so every tile is traced exactly once whatever the hints say; a camera that moves only makes them stale.
#pragma once

namespace rs {

// A hint = { count, effective threshold, list[capacity] } + one flag byte per tile.  Three hints rotate per launch site and stream:
// the launch reads hint `rot`, writes hint `rot + 1` and clears the count of hint `rot + 2` (the out-hint of the launch after it), so
// no launch needs a memset in front of it.  The effective threshold doubles when a launch found more heavy tiles than the list
// holds (who gets a slot is first come, first served, so an overflowing list would leave the longest chains to chance) and falls
// back towards the configured one when the list is mostly empty.
// Everything but three dwords lives in device memory behind `base` (written once per geometry by k_tile_split_init): the walk
// kernels are held to 64 registers, and a struct of seven pointers kept live across the walk cost them 44 bytes of scratch per lane
// and half their speed (k_primary 0.284 -> 0.437 ms) -- the fields are read where they are used instead.
//   ints at base: [0] capacity  [1] configured threshold  [2] number of tiles  [3] ints per hint (2 + capacity)
//                 [4] byte offset of the first flag array  [5] bytes per flag array  [6..7] address of the host's report word (or 0);
//                 hint h at int 8 + h * [3]
// The report word (pinned host memory, written by the last block with one 8-byte store): sequence number of the reporting launch in the
// high half, in the low half the number of heavy tiles the launch BEFORE it found.  The host never waits for it; it reads it when it
// prepares a later launch and lets a launch site whose tiles are all light run the plain kernels for a while (rs_tile_split_prepare).
struct TileSplit {                  // base == null: feature off (every wave is a regular wave)
    int* base;
    int rot;                        // the hint this launch reads, + 3 * the launch's sequence number (every use is modulo 3)
    int helperBlocks;               // blocks at the front of the grid that take listed tiles (= capacity)
};

#if defined(__HIPCC__)
__device__ __forceinline__ int* tile_split_hint(int* base, int h) { return base + 8 + (h % 3) * base[3]; }
__device__ __forceinline__ unsigned char* tile_split_flags(int* base, int h) { return reinterpret_cast<unsigned char*>(base) + base[4] + (h % 3) * base[5]; }

// Which pixels this wave takes.  TW x TH: the regular tile of a wave in pixels (8x8; 8x4 for the two-rays-per-pixel launch, whose
// lanes 32-63 repeat the pixels of lanes 0-31); l = lane index within the pixel set (lane, or lane & 31 for the two-ray launch).
// Returns false when the whole BLOCK has nothing to do (a helper block without a tile) -- block-uniform, so the caller may return
// before any barrier.  Outputs: px, py relative to the launch's first row; active = this lane has a pixel; tile = the tile it
// belongs to (wave-uniform); helper = this block is a helper block.
template <int TW, int TH>
__device__ __forceinline__ bool tile_split_map(const TileSplit& ts, int tilesX, int l, int& px, int& py, bool& active, int& tile, bool& helper) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int perRow = tilesX * 4;
    helper = (int)blockIdx.x < ts.helperBlocks;
    if (helper) {
        const int* in = tile_split_hint(ts.base, ts.rot);
        const int n = min(in[0], ts.helperBlocks);
        tile = (int)blockIdx.x < n ? in[2 + blockIdx.x] : -1;
        if (tile < 0 || tile >= ts.base[2]) return false;
        const int tx = tile % perRow, ty = tile / perRow;
        constexpr int QW = TW / 2, QH = TH / 2;
        px = tx * TW + (wave & 1) * QW + (l % QW);
        py = ty * TH + (wave >> 1) * QH + (l / QW) % QH;
        active = l < QW * QH;
        return true;
    }
    const int b = (int)blockIdx.x - ts.helperBlocks;
    const int bx = b % tilesX, by = b / tilesX;
    tile = by * perRow + bx * 4 + wave;
    px = bx * (TW * 4) + wave * TW + (l % TW);
    py = by * TH + l / TW;
    active = !(ts.base && tile_split_flags(ts.base, ts.rot)[tile]);            // a flagged tile is traced by its helper block
    return true;
}

// after the walk; every lane of the wave calls it (helper blocks: every wave of the block, there is a barrier inside)
__device__ __forceinline__ void tile_split_report(int* base, int rot, int tile, bool helper, bool skipped, unsigned unionNodes) {
    if (!base) return;
    const int lane = threadIdx.x & 63;
    // once per launch, by the last block of the grid (always a regular block, and every wave of a regular block gets here): the next
    // launch's effective threshold, and the count it will add to cleared.  (At the top of the kernel the same few lines made the
    // compiler spill: 64 registers + 44 bytes of scratch per lane instead of 46 registers.)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int* in = tile_split_hint(base, rot);
        tile_split_hint(base, rot + 2)[0] = 0;
        const int cap = base[0], found = in[0], thr = max(in[1], base[1]);
        tile_split_hint(base, rot + 1)[1] = found > cap ? min(thr * 2, 1 << 20) : (found < cap / 4 ? max(base[1], thr - thr / 4) : thr);
        volatile unsigned long long* report = *reinterpret_cast<volatile unsigned long long**>(base + 6);
        if (report) *report = ((unsigned long long)(unsigned)(rot / 3) << 32) | (unsigned)found;
    }
    if (helper) {
        __shared__ unsigned quadrant[4];
        if (lane == 0) quadrant[threadIdx.x >> 6] = unionNodes;
        __syncthreads();
        if (threadIdx.x != 0) return;
        unionNodes = quadrant[0] + quadrant[1] + quadrant[2] + quadrant[3];       // >= the undivided tile's count
    }
    else if (lane != 0 || skipped) return;
    int* out = tile_split_hint(base, rot + 1);
    const unsigned threshold = (unsigned)max(tile_split_hint(base, rot)[1], base[1]);        // the effective threshold of this launch
    bool keep = false;
    if (unionNodes >= threshold) { const int slot = atomicAdd(out, 1); keep = slot < base[0]; if (keep) out[2 + slot] = tile; }
    tile_split_flags(base, rot + 1)[tile] = keep ? 1 : 0;
}

// once per geometry: header and empty hints (the arrays behind them have been zeroed by a memset on the same stream)
static __global__ void k_tile_split_init(int* base, int capacity, int threshold, int numTiles, int flagOffset, int flagStride, unsigned long long* report) {
    base[0] = capacity; base[1] = threshold; base[2] = numTiles; base[3] = 2 + capacity; base[4] = flagOffset; base[5] = flagStride;
    *reinterpret_cast<unsigned long long**>(base + 6) = report;
}
#endif

}  // namespace rs
