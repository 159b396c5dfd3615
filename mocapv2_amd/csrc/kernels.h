// kernels.h -- internal launch interface between the C-ABI (abi.hip) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mocap {

// The dense filter kernel (any geometry, any lens model): every tile of every image, the undistortion as a gather in the
// row pipeline.  Used when every tile has to be filtered anyway (early-out off or not provable), for images narrower than
// 8 pixels, for undistort tables whose displacements do not fit the compact table of the box kernel, and by the
// single-image convenience entry points.
struct FilterArgs {
    const uint8_t* src;   // images, image_stride bytes apart, rows `pitch` bytes apart
    size_t image_stride;
    int pitch, H, W;
    uint32_t* mask;       // [n_images][H][words_per_row], bit b of word k = pixel 32k+b
    int words_per_row;
    uint32_t* cells;      // [n_images][n_cgroups*4][n_strips]: bit g = rows 8g..8g+7 of the chunk have set pixels in the strip; bit 31 = tile ran the full filter
    // undistort tables of the first slot used (remap variant only), each [cam_mod][H][W]:
    const uint32_t* map;  //   tap position: (sx - x) | (sy - y) << 16, the 2x2 tap window clamped into the image
    const uint32_t* mapw; //   blend weights 32*(wx0 | wx1<<8 | wy1<<16 | wy0<<24)/32, taps outside the image weigh 0
    const uint4* tiles; const uint32_t* n_tiles; uint32_t cap_tiles; // list form (launch_filter_tiles): entries image, tile (chunk *
                          //   n_strips + strip), first row, last row; their number
    int pipelined;        // 1 = software-pipelined gather (table words 8 rows ahead, taps 4 rows ahead; W % 4 == 0, H >= 2), 0 = per-pixel gather
    int cam_mod;          // undistort slot of image n = n % cam_mod (map already points at the first slot)
    int n_images, n_steps; // n_steps = ceil(n_images / cam_mod)
    int thr_mul;          // floor(thresh)+1: blurred > thresh  <=>  S >= thr_mul * taps
    int rows_per_chunk, n_strips, n_cgroups;
    // the staged form of the row pipeline (filter_rows_staged_kernel): compact table, source pixels through LDS
    int staged = 0;       // 1 = use it (remap variant; W % 16 == 0, H >= 2, every slot's table compact)
    const uint32_t* map4 = nullptr; // compact table of the first slot used (see BoxArgs)
    const ushort4* rowbox = nullptr; // [cam_mod][H][n_strips]: per row and strip, the box of tap coordinates the strip's pixels read, + 2
    int stage_dw = 0;         // dwords of LDS a band's source rectangle may take (ROWS_STAGE_DW; smaller values are a test switch)
};
void launch_rowbox(const uint32_t* map4, ushort4* rowbox, int H, int W, int n_strips, hipStream_t s);
int rows_stage_dwords();

// The sparse half of the filter stage (blob_boxes.hip): tiles, boxes, work items.
// A tile = 240 mask columns x rows_per_chunk rows.  The scan kernel leaves per tile the box of mask rows / columns that
// hot cells can reach (tile_rows); settle_tiles_kernel turns the boxes into items; box_filter_kernel consumes them.
#ifndef MOCAP_BOX_HCAP         // (build-time knobs for A/B builds, scratch/build_variant.sh)
#define MOCAP_BOX_HCAP 1536
#define MOCAP_BOX_SCAP 6656
#define MOCAP_BOX_MAX_PARTS 4
#endif
constexpr int BOX_HCAP = MOCAP_BOX_HCAP; // quad-rows (4 pixels x 1 row) of one item's patch (halving it for twice the waves per CU: 0.63 against 0.53 ms)
constexpr int BOX_SCAP = MOCAP_BOX_SCAP; // bytes of source pixels staged in LDS per item (>= (BOX_HCAP + 64) * 4: the counts alias it)
constexpr int BOX_MAX_PARTS = MOCAP_BOX_MAX_PARTS;  // items a tile is cut into at most (a whole 240 x 68 tile: 2 x 2)
struct BoxItem { uint32_t image, tile, x01, y01, bx01, by01, pad0, pad1; }; // output region: columns x0 | x1 << 16, rows y0 | y1 << 16
                                                                        //   (x0 > x1: skip); the scan's box of the tile (not clipped to it)
struct BoxArgs {
    const uint8_t* src; size_t image_stride; int pitch, H, W;
    uint32_t* mask; int words_per_row;
    uint32_t* cells;            // occupancy words [n_images][n_chunks][n_strips] (see FilterArgs)
    const uint32_t* map4;       // compact undistort table of the first slot used, [cam_mod][H][W]: dx (11 bits, signed) |
                                //   dy (11, signed) << 11 | x fraction (5) << 22 | y fraction (5) << 27; the 2x2 taps start at
                                //   (x + dx, y + dy), clamped into [-2, W] x [-2, H] (taps outside the image read 0)
    const ushort4* srcbox;      // [cam_mod][ceil(H/8)][ceil(W/8)]: box of the tap coordinates of an 8x8 output cell, + 2
    uint64_t remap_bits;        // bit s: slot s (relative to the first) is remapped (else the identity)
    int cam_mod, n_images, n_steps;
    int thr_mul;
    int rows_per_chunk, n_strips, n_chunks;
    uint32_t* tile_rows;        // [n_images][n_chunks][n_strips][4]: the scan's boxes of this batch (first / last row, first / last
                                //   column; (0xffffffff, 0) = none), read-only for settle
    uint32_t* tile_rows_next;   // the same array for the next batch (the two alternate): emptied by settle
    int n_clear;                // images whose boxes tile_rows_next may still hold (an earlier, larger batch): emptied too
    int cluster;                // 1 = a box spanning tiles that all hold it becomes one item (A/B switch)
    uint32_t* cur_box;          // [n_images][n_chunks][n_strips][4]: words 0-1 the tile's output region of this batch
                                //   (x0 | x1 << 16, y0 | y1 << 16; x0 > x1 = none) = what the mask may hold there; words 2-3 the
                                //   scan's box (not clipped to the tile)
    BoxItem* items; uint32_t* n_items; uint32_t cap_items; // n_items[0] = items in the list; n_items[16 + 16 x] = head of run x (box kernel)
    uint64_t* timing;           // optional [grid][6] phase clock of the box kernel (MOCAP_BOX_TIMING=1, a debugging aid), else null
    int prio;                   // 1 = raise the wave priority of the box kernel (A/B switch)
    int stage_bytes;            // LDS bytes the source staging may use (BOX_SCAP; smaller values are a test switch)
    uint4* wide_tiles;          // tiles whose box is wider than `wide_quads` patch quads go to this list instead (image, tile, first row,
                                //   last row; counted in n_items[8]) and through the sliding row pipeline of filter_mask_kernel, which
                                //   does less work per pixel on wide regions than the box kernel
    uint32_t cap_wide; int wide_quads_remap, wide_quads_identity, wide_bands;
    uint32_t* zero8;            // not null: 8 words (the contour stage's walk counters) that settle zeroes -- in place of a fill launch
    int dense;                  // 1 = no early-out: every tile is filtered whole
    int ext_mask;               // 1 = caller-owned mask (cleared by the scan kernel, or written whole when dense)
};
void launch_settle_tiles(const BoxArgs& a, hipStream_t s);
void launch_box_filter(const BoxArgs& a, int grid, hipStream_t s);
int box_filter_blocks_per_cu();
void launch_srcbox(const uint32_t* map4, ushort4* srcbox, int H, int W, hipStream_t s);

struct MapArgs {
    double K[9], dist[5];
    int H, W;
    uint32_t* map;   // [H][W] tap positions (see FilterArgs)
    uint32_t* mapw;  // [H][W] blend weights
    uint32_t* map4;  // [H][W] compact table (see BoxArgs)
    uint32_t* flags; // bit0: the table is not the identity; bit1: a displacement does not fit the compact table
};

// one border found by the contour kernel (also the debug record compared with the oracle in tests)
struct ContourRec {
    int32_t key;       // raster index (y*(W+1)+x) of the scan position that discovers the border
    int32_t is_hole;
    int32_t sx, sy;    // first border pixel
    int32_t npts;      // CHAIN_APPROX_SIMPLE vertex count
    int32_t steps;
    int64_t a00, a10, a01;
    double area, perimeter;
    int32_t kept, cx, cy;
    int32_t link;      // rec index of the border owning the crack left of the start (-1 frame), see kernel
    int32_t parent;    // rec index of the parent border, -1 = frame
    int32_t order;     // position among the kept contours in cv.findContours order, -1 if not kept
};

struct ContourArgs {
    const uint32_t* mask;
    int words_per_row, H, W, n_images;
    int32_t* out_xy;     // image n: out_xy + n*xy_stride, [max_blobs][2]
    int32_t* out_count;  // image n: out_count[n*count_stride]; may exceed max_blobs (truncated), <0 = error
    long xy_stride, count_stride;
    int max_blobs;
    double min_area, min_circ;
    ContourRec* dbg;     // optional [n_images][dbg_cap]
    int32_t* dbg_count;  // optional [n_images]
    int dbg_cap;
    int max_steps;
    const uint32_t* cells; // occupancy written by the filter kernel (see FilterArgs), or null = scan every row
    const uint32_t* boxes; // with cells: per tile [4] the output region and the scan's box (BoxArgs::cur_box), or null = whole strips
    int rows_per_chunk, n_chunks, n_strips;
    uint64_t* timing;      // optional [n_images][8] phase clock (debugging aid), else null
    void* work;            // [n_images] per-image workspace of contour_work_bytes() each
    int prio;              // wave priority (s_setprio 0..3): the walks are serial chains, cheap to favour and costly to delay
    uint64_t* walk_list;   // split form: [n_images * contour_walk_bytes() / 8] candidate walks of the batch (blob_contours.hip: walk_entry), or
                           //   null = one kernel per image
    uint64_t* link_list;   // split form: [n_images * contour_link_bytes() / 8] link walks of the batch (second follow pass)
    uint32_t* walk_count;  // split form: [8] entries in walk_list, its head, entries in link_list, its head, entries in wait_list
                           //   (zeroed by launch_contours)
    uint32_t* wait_list;   // split form: [n_images] the images the first tree pass left to the second one
    int follow_grid;       // split form: workgroups (waves) of the follow kernel
    int follow_grid2;      //   ... of its second pass (the link walks: few)
    int counters_zeroed;   // split form: 1 = walk_count was zeroed by an earlier kernel of the stream (settle): no fill launch
    int defer_links;       // split form: 1 = links only a walk can settle go through a second follow + tree pass (two more launches);
                           //   0 = the first tree pass walks them in place (one wave per link) and the second passes are not launched
    int image_grid;        // split form: > 0 = the per-image kernels (candidates, tree) as that many workgroups looping over the images
    int follow_list;       // set by launch_contours: 0 = the follow kernel works through walk_list, 1 = through link_list
    int tree_pass;         // set by launch_contours: 1 / 2 = first / second pass of the tree kernel
    uint64_t* follow_dbg;  // optional [follow_grid][8] phase clock of the follow kernel (follow_timing = 1: first pass, 2: second), else null
    int follow_dbg_list;
};

enum { BLOB_ERR_CANDIDATES = -2, BLOB_ERR_CONTOURS = -3, BLOB_ERR_STEPS = -4, BLOB_ERR_DEPTH = -5 };

void launch_filter_mask(const FilterArgs& a, bool remap, hipStream_t s);
void launch_filter_tiles(const FilterArgs& a, bool remap, int blocks, hipStream_t s);
constexpr int PROBE_STRIDE = 32; // words between two pairs of probe counters: a cache line each
struct BrightArgs {
    const uint8_t* src; size_t image_stride; int pitch, H, W, n_images; // W >= 8
    int cam_mod;                  // undistort slot of image n = n % cam_mod (the tables already point at the first slot)
    uint32_t ncx_magic;           // ceil(2^32 / d), d = ceil(W/8) (wide: d / 2), if that divides every index exactly by multiply-high, else 0
    int wide;                     // 1 = 16-byte loads (W, pitch, image stride, base all multiples of 16; ncx_magic != 0)
    int base;                     // excess base c: a pixel p counts with max(0, p - c)
    int hot, hot_edge, hot_corner; // a cell whose doubled excess sum (2 * sum of max(0, p - c)) exceeds this is hot (4 * hot <= allow); _edge /
                                  //   _corner for cells feeding windows the image border cuts in one axis / in both
    const uint2* reach;           // [cam_mod][cells]: box of the output pixels that read the 8x8 source cell, x0 | x1 << 16, y0 | y1 << 16
    const uint8_t* cflags;        // [cam_mod][cells]: 1 / 2 = the cell feeds windows the image border cuts in one axis / in both
    uint32_t* tile_rows; int n_chunks, n_strips; uint32_t rows_magic; // reachable mask rows / columns per tile, see FilterArgs;
                                  //   rows_magic = ceil(2^23 / rows per chunk)
    uint32_t* mask; size_t mask_words; int mask_aligned16; // caller-owned bit masks to clear on the side (mask_words = 0: none)
    // probe (optional): on every 16th image also count the cells that are hot under the current base and under the alternative
    // base `base_alt` / threshold `hot_alt`, into probe[PROBE_STRIDE * (block & 127) + 0 / 1]: the host compares the sums and switches
    uint32_t* probe; int base_alt, hot_alt;
    int prio;                     // wave priority of the scan (s_setprio): its few instructions are loads that keep HBM busy
    int max_blocks;               // > 0: launch at most this many workgroups, each looping over the batch's blocks (persistent form)
    uint32_t* block_ctr;          // persistent form: [slices] zeroed counters the workgroups take their blocks from (else null: fixed stride)
    int blocks_x;                 // blocks of 256 threads per image (set by launch_bright_cells)
    int slices;                   // > 1: the pass goes out as that many launches over consecutive runs of images
    int image0, slice_images;     // set by launch_bright_cells: the images of this launch
    uint32_t* hotmap;             // [n_images][hot_words]: two bits per source cell, cell i of an image at bits 2 (i % 16) of word i / 16: how
                                  //   many of the thresholds hot_corner <= hot_edge <= hot its sum exceeds.  All zeros between batches: the scan
                                  //   stores the words that are not zero, mark_tiles_kernel reads and clears them.  null = the scan marks the tiles itself
    int hot_words;                // words per image: hot_map_words(H, W, wide)
    uint32_t* zero_counters;      // not null: 256 words (the context's counter block) that workgroup 0 of the scan zeroes -- in place of a fill launch
    int mark_grid;                // > 0: workgroups of mark_tiles_kernel (its waves loop over the map); 0 = one piece per wave
};
int hot_map_words(int H, int W, int wide);
void launch_mark_tiles(const BrightArgs& a, hipStream_t s); // hot map -> tile boxes (tile_rows), behind launch_bright_cells
void launch_bright_cells(const BrightArgs& a, hipStream_t s);
// set-up statistics of an undistort table, for the dark-tile bound: stats[0] = largest total blend weight any source
// pixel carries over all output pixels (1024 = one full pixel), stats[1] / stats[2] = largest x / y extent (in source
// pixels) of the taps feeding one 5x5 output window.  acc: H*W zero-initialised scratch words.  edge: zero-initialised
// [ceil(H/8)][ceil(W/8)] words, bit 0 / bit 1 set for the 8x8 source cells read by windows that the image border cuts
// in exactly one axis / in both.  reach: [cells][4] = x0, x1, y0, y1 initialised to (INT_MAX, INT_MIN, INT_MAX, INT_MIN):
// bounding box of the output pixels that read the cell with a nonzero weight.
struct StatArgs { const uint32_t* map; const uint32_t* mapw; uint32_t* acc; uint32_t* stats; int H, W; uint32_t* edge; int* reach; };
void launch_remap_stats(const StatArgs& a, hipStream_t s);
void launch_undistort_map(const MapArgs& m, hipStream_t s);
void launch_contours(const ContourArgs& a, hipStream_t s);
size_t contour_work_bytes();
size_t contour_walk_bytes(); // walk list bytes per image (split form of the contour stage)
size_t contour_link_bytes(); // link list bytes per image
void launch_box_blur(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, int ksize, hipStream_t s);
void launch_undistort(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, const uint32_t* map, const uint32_t* mapw,
                      hipStream_t s);
void launch_mask_expand(const uint32_t* mask, int wpr, uint8_t* dst, int H, int W, int dp, hipStream_t s);
void launch_median5(const uint8_t* src, uint8_t* dst, int H, int W, int sp, int dp, int ithresh, int apply, hipStream_t s);
void launch_demosaic(const uint8_t* bayer, uint8_t* bgr, int H, int W, int sp, hipStream_t s);
// Bayer -> gray (bayer_gray.hip): red sites at row parity ry / column parity rx, luma coefficients cb, cg, cr >> shift
struct BayerArgs {
    const uint8_t* src; uint8_t* dst;
    int H, W, n_images;
    long spitch, dpitch;
    size_t sstride, dstride;
    int ry, rx;
    uint32_t cb, cg, cr;
    int shift;
};
void launch_bayer_gray(const BayerArgs& a, hipStream_t s);
// the same pass fused with the early-out's scan of the gray frames it writes (bayer_scan_fusable: W % 16 == 0, H % 8 == 0, aligned)
bool bayer_scan_fusable(const BayerArgs& a);
void launch_bayer_gray_scan(const BayerArgs& a, const BrightArgs& b, hipStream_t s);

// ---- geometry ----
struct CameraTable {           // device-resident, written by mocap_set_cameras / mocap_set_fundamentals
    double K[32][9], dist[32][5], R[32][9], t[32][3];
    double F[31][9];           // F[i-1]: camera-0 pixel -> epipolar line in camera i
    int n_cam, n_F;
};

struct CorrArgs {
    const CameraTable* cams;
    const void* pts;           // points of (t,c): pts + (t*pt_st + c*pt_sc) elements, [P][2] int32 or float64
    const int32_t* counts;     // count of (t,c): counts[t*cnt_st + c*cnt_sc]
    long pt_st, pt_sc, cnt_st, cnt_sc;
    int pts_f64;
    int T, C, P;               // P = capacity per camera
    double cutoff;
    int max_groups;            // per root
    // per time step, per camera-0 root
    double* root_xyz;          // [T][P][3]
    double* root_err;          // [T][P]  mean reprojection error over the root's groups
    double* root_grp;          // [T][P][C][2] first group
    int32_t* root_idx;         // [T][P]  camera-0 index of the j-th surviving root
    int32_t* order;            // [T][P]  argsort(root_err)
    int32_t* n_roots;          // [T]  (<0 = error)
    int prio;                  // wave priority (s_setprio 0..3), see ContourArgs
    int threads;               // threads per time step: 64 / 128 / 256 (A/B switch; one wave per step measured 0.064 against 0.051 ms)
    double* scratch;           // [T][step_budget] per-group errors, the groups of a time step back to back in root order
    int step_budget;           // groups per time step the scratch holds
    int lds_budget;            // set by launch_correspond: bytes of dynamic LDS the kernel's plan is made for
};

struct TriArgs {
    const CameraTable* cams;
    const double* pts;         // [N][C][2]
    const uint8_t* valid;      // [N][C]
    int N, C;
    int compact_k;             // intrinsics indexed by position after dropping invalid entries (reference quirk)
    double* xyz;               // [N][3]
    int32_t* ok;               // [N] 1 = triangulated
};

struct ReprojArgs {
    const CameraTable* cams;
    const double* pts;         // [N][C][2]
    const uint8_t* valid;      // [N][C]
    const double* xyz;         // [N][3]
    int N, C, compact_k;
    double* mse;               // [N]
    int32_t* ok;               // [N]
};

struct EpiArgs {
    const CameraTable* cams;
    const void* roots;         // [n_roots][2] camera-0 points (int32 or float64)
    const void* cand;          // [n_cand][2] points of the other camera
    int n_roots, n_cand, pts_f64, f_index;
    double* dist;              // [n_roots][n_cand]
    float* lines;              // optional [n_roots][3]
};

struct BaArgs {
    const CameraTable* cams;   // K and dist of cameras 0..C-1 (R, t come from the parameters)
    const double* params;      // [B][6 (C - 1)] rotvec + t of cameras 1..C-1, device-accessible (pinned host memory is fine)
    const double* pts;         // [N][C][2]
    const uint8_t* valid;      // [N][C]
    int N, C, B;
    double* obj;               // scratch [B][N][3]
    float* res;                // [B][N], the first counts[b] entries of row b are the residual vector
    int32_t* counts;           // [B]
};

enum { CORR_ERR_GROUPS = -2, CORR_ERR_TRUNCATED = -3, CORR_ERR_BLOB = -4 };

void launch_correspond(const CorrArgs& a, hipStream_t s);
size_t correspond_smem_bytes(int P, int C);
bool correspond_fits(int P, int C); // the kernel's LDS plan for P points x C cameras fits a workgroup
void launch_epipolar_scores(const EpiArgs& a, hipStream_t s);
void launch_ba_residuals(const BaArgs& a, hipStream_t s);
void launch_triangulate(const TriArgs& a, hipStream_t s);
void launch_reproject(const ReprojArgs& a, hipStream_t s);

} // namespace mocap
