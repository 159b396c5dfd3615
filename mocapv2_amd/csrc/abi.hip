// abi.hip -- the C-ABI of libmocap_hip.so (declared in include/mocap_hip.h): argument checks, the per-GPU
// context (undistort tables, camera table, scratch), kernel launches.  No compute happens on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <mutex>
#include <memory>
#include <dlfcn.h>
#include "../../include/mocap_hip.h"
#include "kernels.h"

using namespace mocap;

static_assert(sizeof(mocap_contour) == sizeof(ContourRec), "debug record layout");

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return fail(MOCAP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

struct EvPair { hipEvent_t a, b; };
// the scan's probe counters (BrightArgs::probe): 128 pairs, each pair in a cache line of its own (PROBE_STRIDE words apart) -- packed
// into 8 lines, the ~200 k atomic adds of a probed batch queued up on 8 L2 atomic units: 0.3 ms on top of the scan's 0.95
constexpr size_t PROBE_BYTES = 128 * mocap::PROBE_STRIDE * sizeof(uint32_t);
using mocap::PROBE_STRIDE;

// One RCCL communicator per rank, shared by the rank's contexts (= the batches in flight): RCCL wants the operations of a
// communicator issued one after the other, so every all-gather waits for the event its predecessor recorded (on whatever
// stream that one ran) and records it anew.
struct SharedComm {
    void* comm = nullptr;      // ncclComm_t
    int rank = 0, world = 1, device = 0;
    hipEvent_t last = nullptr; // completion of the most recent all-gather on this communicator
    bool have_last = false;
    std::mutex mu;             // issue order = lock order
    bool (*destroy_comm)(void*) = nullptr; // ncclCommDestroy, bound when the communicator is created
    // the last context sharing the communicator is gone (mocap_comm_destroy, or a context destroyed without it): nothing leaks
    ~SharedComm()
    {
        if (!comm && !last) return;
        (void)hipSetDevice(device);
        if (last) { if (have_last) (void)hipEventSynchronize(last); (void)hipEventDestroy(last); }
        if (comm && destroy_comm) (void)destroy_comm(comm);
    }
};

// Performance switches of a context (A/B measurements, tests of the alternative code paths; none changes a result).  They are
// read from the environment ONCE, by mocap_ctx_create (MOCAP_<NAME IN CAPITALS>), and can be changed per context afterwards
// with mocap_set_tuning: the hot path itself never calls getenv.  -1 = "not set" where 0 is a meaningful value.
struct Tuning {
    int rows = 68;               // rows per tile, 16..68
    int general_filter = 0;      // 1 = the dense kernel for everything
    int skip_dark = 1;           // 0 = no early-out: every tile is filtered
    int dense_boxes = 0;         // 1 = with skip_dark = 0: the box kernel on whole tiles instead of the dense kernel
    int remap_pipeline = 1;      // 0 = the dense kernel's per-pixel gather
    int cluster = 1;             // 0 = no 2x2-tile cluster items
    int wide_quads_remap = 34, wide_quads_identity = 34; // routing threshold box kernel / row pipeline (1000 = never); round 4: 40 -> 34 with three
                                 //   workgroups of the row pipeline per CU (filters 0.49 -> 0.47 ms at 8 markers, 1.25 -> 1.19 at 32)
    int wide_bands = 1;          // row bands of a wide tile, 1..4
    int wide_fork = 0;           // 1 = the wide tiles on a side stream beside the box kernel
    int box_prio = 0, scan_prio = 0, contour_prio = 0, corr_prio = 0; // wave priorities
    int box_stage_bytes = BOX_SCAP; // LDS the box kernel's source staging may use (0 = taps from memory)
    int box_blocks_per_cu = 0;   // 0 = box_filter_blocks_per_cu()
    int box_timing = 0, contour_timing = 0, follow_timing = 0; // phase clocks on stderr (synchronous debugging aids)
    int scan_wide = 1;           // 0 = the scan's 8-byte loads
    int rows_staged = 0;         // 1 = the row pipeline (dense path, wide tiles) on the compact table with its source pixels staged in LDS instead of the
                                 //   general 8-byte tables with tap gathers (measured: 0.35 against 0.18 ms for the wide tiles of the benchmark batch -- the
                                 //   4-byte table costs ~24 instead of ~14 VALU instructions per pixel and the stage is issue-bound: not the default)
    int wide_blocks_per_cu = 3;  // workgroups (4 waves) of the wide-tile kernel per CU (137 registers: three waves per SIMD)
    int rows_stage_dw = -1;      // dwords of LDS a band's source rectangle may take (-1 = all of the buffer; 0 = taps from memory: a test switch)
    int scan_serial = 1;         // 1 = one streaming scan at a time on the device: a context's scan waits for the scan launched before it (event chain across
                                 //   contexts).  A scan alone saturates HBM; two at once only stretch each other and the chains behind them (round 4: +5..9 %)
    int scan_hotmap = 1;         // who turns hot cells into tile boxes: 0 = the scan itself (reach lookup + atomics behind its loads), 2 = mark_tiles_kernel
                                 //   from the hot map the scan leaves, 1 = whichever the last probe's hot-cell count favours (crowded scenes: the map)
    int scan_blocks_per_cu = 0;  // > 0: the scan as a persistent pass of that many workgroups per CU (0 = one workgroup per block)
    int scan_slices = 1;         // the scan goes out as that many launches over consecutive runs of images
    int excess_base = -1;        // >= 0: pins the scan's excess base
    int base_sel = 1;            // the base a context starts with (0 tight, 1 tolerant)
    int probe_debug = 0;
    int contour_boxes = 1;       // 0 = candidates from whole strips
    int contours_split = 1;      // 0 = the contour stage as one kernel per image
    int contour_defer = 1;       // links that need a walk: 2 = always through the second follow / tree passes, 0 = walked in place, 1 = by the scene
    int mark_blocks_per_cu = 0;  // > 0: mark_tiles_kernel as that many workgroups per CU whose waves loop over the hot map
    int contour_blocks_per_cu = 0; // > 0: the per-image contour kernels (candidates, tree) as that many workgroups per CU looping over the images
    int corr_threads = 256;      // threads per time step of the correspondence kernel (64 / 128 / 256)
    int corr_step_groups = 0;    // candidate groups one time step may hold in all (error scratch per step); 0 = max(2 * max_groups, 8192)
};
struct TuneName { const char* name; int Tuning::*field; int lo, hi; };
static const TuneName kTuneNames[] = {
    {"rows", &Tuning::rows, 16, 68}, {"general_filter", &Tuning::general_filter, 0, 1}, {"skip_dark", &Tuning::skip_dark, 0, 1},
    {"dense_boxes", &Tuning::dense_boxes, 0, 1}, {"remap_pipeline", &Tuning::remap_pipeline, 0, 1}, {"cluster", &Tuning::cluster, 0, 1},
    {"wide_quads_remap", &Tuning::wide_quads_remap, 0, 100000}, {"wide_quads_identity", &Tuning::wide_quads_identity, 0, 100000},
    {"wide_bands", &Tuning::wide_bands, 1, 4}, {"wide_fork", &Tuning::wide_fork, 0, 1},
    {"box_prio", &Tuning::box_prio, 0, 1}, {"scan_prio", &Tuning::scan_prio, 0, 3}, {"contour_prio", &Tuning::contour_prio, 0, 3},
    {"corr_prio", &Tuning::corr_prio, 0, 3}, {"box_stage_bytes", &Tuning::box_stage_bytes, 0, BOX_SCAP},
    {"box_blocks_per_cu", &Tuning::box_blocks_per_cu, 0, 32}, {"box_timing", &Tuning::box_timing, 0, 1},
    {"contour_timing", &Tuning::contour_timing, 0, 1}, {"follow_timing", &Tuning::follow_timing, 0, 2}, {"scan_wide", &Tuning::scan_wide, 0, 1}, {"scan_hotmap", &Tuning::scan_hotmap, 0, 2}, {"scan_serial", &Tuning::scan_serial, 0, 1}, {"rows_staged", &Tuning::rows_staged, 0, 1}, {"wide_blocks_per_cu", &Tuning::wide_blocks_per_cu, 1, 8}, {"rows_stage_dw", &Tuning::rows_stage_dw, -1, 1 << 20}, {"scan_blocks_per_cu", &Tuning::scan_blocks_per_cu, 0, 64}, {"scan_slices", &Tuning::scan_slices, 1, 64}, {"excess_base", &Tuning::excess_base, -1, 254},
    {"base_sel", &Tuning::base_sel, 0, 1}, {"probe_debug", &Tuning::probe_debug, 0, 1}, {"contour_boxes", &Tuning::contour_boxes, 0, 1},
    {"contours_split", &Tuning::contours_split, 0, 1}, {"contour_blocks_per_cu", &Tuning::contour_blocks_per_cu, 0, 16}, {"mark_blocks_per_cu", &Tuning::mark_blocks_per_cu, 0, 16}, {"contour_defer", &Tuning::contour_defer, 0, 2}, {"corr_threads", &Tuning::corr_threads, 64, 256},
    {"corr_step_groups", &Tuning::corr_step_groups, 0, 0x7fffffff},
};
static bool tune_set(Tuning& t, const char* name, int v)
{
    for (const TuneName& n : kTuneNames)
        if (!strcmp(n.name, name)) {
            t.*(n.field) = v < n.lo ? n.lo : (v > n.hi ? n.hi : v);
            return true;
        }
    return false;
}
static Tuning tuning_from_env()
{
    Tuning t;
    for (const TuneName& n : kTuneNames) {
        char env[64] = "MOCAP_";
        size_t k = strlen(env);
        for (const char* p = n.name; *p && k + 1 < sizeof(env); p++) env[k++] = (char)((*p >= 'a' && *p <= 'z') ? *p - 32 : *p);
        env[k] = 0;
        const char* e = getenv(env);
        if (e && *e) tune_set(t, n.name, atoi(e));
    }
    { const char* e = getenv("MOCAP_WIDE_QUADS"); int r_ = 0, i_ = 0; // "remap,identity"
      if (e && sscanf(e, "%d,%d", &r_, &i_) == 2) { tune_set(t, "wide_quads_remap", r_); tune_set(t, "wide_quads_identity", i_); } }
    return t;
}

struct mocap_ctx {
    int device, W, H, n_slots, wpr;
    int box_grid;             // workgroups of the box kernel: the resident ones (box_filter_blocks_per_cu() per CU)
    int n_cu;                 // compute units of the device
    mocap_blob_params prm;
    Tuning tune;
    uint32_t* maps;           // [2][n_slots][H][W]: tap positions, then blend weights (general form)
    uint32_t* map4;           // [n_slots][H][W] (+ 4 words): compact table of the box kernel
    ushort4* srcbox;          // [n_slots][ceil(H/8)][ceil(W/8)]: source box per 8x8 output cell (box kernel)
    ushort4* rowbox;          // [n_slots][H][n_strips]: source box per row and strip (staged row pipeline)
    uint32_t* map_flags;      // [n_slots] device
    std::vector<int> slot_state; // 0 unset, 1 identity, 2 remap
    std::vector<int> slot_compact; // 1 = the slot's displacements fit the compact table (identity: always)
    std::vector<uint32_t> slot_wmax; // largest total blend weight of a source pixel (1024 = identity); 0 = early-out not provable
    uint2* reach;             // [n_slots][ceil(H/8)][ceil(W/8)] per 8x8 source cell: box of the output pixels that read it
    uint8_t* cflags;          // [n_slots][cells] border-cut window flags per source cell (see BrightArgs)
    uint32_t* mask; size_t mask_images;
    bool mask_dirty;                       // the general kernel wrote the mask whole: clear it before the box path runs again
    uint32_t* cells; size_t cells_images; // occupancy cells written by the filter kernels for c->mask
    int last_images;                       // images of the most recent batch that wrote c->cells
    uint32_t* hotmap;                      // [mask_images][hot_map_words(H, W, 1)] the scan's hot map (BrightArgs::hotmap)
    uint32_t* tile_rows;                   // [2][mask_images][tiles][4] the scan's box per tile (see BoxArgs): two arrays, alternating
    int tile_rows_flip;                    //   per batch: the one the scan widens and settle reads / the one settle empties
    int tile_rows_hold[2];                 //   images whose boxes each of the two may still hold (batches of varying size)
    uint32_t* cur_box;                     // [mask_images][tiles][4] output region / scan box of the last batch per tile (BoxArgs)
    BoxItem* items; uint32_t* n_items; uint32_t cap_items; // work list of the box kernel
    uint4* wide_tiles; uint32_t cap_wide;                  // list of the tiles with wide boxes (filter_mask_kernel, list form)
    hipStream_t side; hipEvent_t ev_fork, ev_join;         // the wide tiles are filtered beside the box kernel: side stream, fork / join events
    // excess base of the scan, adapted between batches: two candidates (tight / tolerant of bright backgrounds), the current
    // one, and a probe now and then that counts the hot cells both would leave (BrightArgs::probe)
    int base_sel; int probe_age; bool probe_pending; uint32_t* probe_dev; uint32_t* probe_host; hipEvent_t probe_ev;
    bool walk_count_zeroed;                // the filter stage of the current batch has zeroed walk_count (settle_tiles_kernel)
    int probe_images;                      // images the pending probe counted on (every 16th of its batch)
    bool hot_dense;                        // the last probe found a crowded scene (many hot cells per image): the scan leaves a hot map
    uint32_t* cells_ext; uint32_t* cur_box_ext; size_t cells_ext_images; // the same for caller-owned masks (mocap_filter_mask)
    void* cwork; size_t cwork_images;      // contour kernel workspace, contour_work_bytes() per image
    uint64_t* walk_list; uint64_t* link_list; uint32_t* walk_count; // contour stage, split form: the batch's border walks / link walks
                                                                    //   (grown with cwork) and their counters
    CameraTable* cams; int n_cam, n_F;
    double* scratch; size_t scratch_elems;
    double* ba_obj; size_t ba_obj_elems;   // object points of mocap_ba_residuals, [B][N][3]
    void* ba_pinned; size_t ba_pinned_bytes; // its host-side hand-over: parameters in, residuals + counts out (the kernel reads / writes it directly)
    std::shared_ptr<struct SharedComm> comm; // RCCL communicator of mocap_comm_init / mocap_comm_share, else null
    bool profiling;
    std::vector<EvPair> ev[5];
    std::mutex mu;
};

static int set_device(mocap_ctx* c) { HIP_TRY(hipSetDevice(c->device)); return 0; }

// scan_serial: the scans of all contexts of a device form one chain (each waits for the completion event of the one launched
// before it), so that two batches' scans never share the chip -- a scan alone saturates HBM, two at once only delay each other
struct ScanTurn { std::mutex mu; hipEvent_t done = nullptr; bool have = false; };
static ScanTurn g_scan_turn[64];

struct Tiling { int rows, n_cgroups, n_strips; };
static Tiling tiling(const mocap_ctx* c)
{
    Tiling t;
    // Rows per tile.  Must be <= 68: settle_tiles_kernel cuts a tile into at most 4 items of BOX_HCAP quad-rows.
    t.rows = c->tune.rows;
    if (c->H < 4 * 32) t.rows = (c->H + 3) / 4 > 8 ? (c->H + 3) / 4 : 8;
    t.n_cgroups = (c->H + 4 * t.rows - 1) / (4 * t.rows);
    t.n_strips = (c->W + 239) / 240;
    return t;
}

static size_t source_cells(const mocap_ctx* c) { return (size_t)((c->H + 7) / 8) * ((c->W + 7) / 8); }
static size_t cells_per_image(const mocap_ctx* c) { Tiling t = tiling(c); return (size_t)t.n_cgroups * 4 * t.n_strips; }

// ---- RCCL, bound at run time -----------------------------------------------------------------------------------
// The path's one exchange is an ncclAllGather of centroid records.  librccl is looked up with dlopen
// when the first communicator call arrives: a process that already holds RCCL (PyTorch ships its own librccl.so.1)
// shares that copy, a plain C host gets /opt/rocm/lib's; single-GPU users never load it.
struct IdBytes { char internal[MOCAP_COMM_ID_BYTES]; }; // layout of ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, IdBytes, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
}
static Rccl g_rccl;
static std::mutex g_rccl_mu;
static int load_rccl()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MOCAP_E_UNSUPPORTED, "librccl.so.1 not found: %s", dlerror());
    Rccl r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString)
        return fail(MOCAP_E_UNSUPPORTED, "librccl lacks an expected symbol");
    r.lib = h;
    g_rccl = r;
    return 0;
}
#define RCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        int r_ = (expr);                                                                                      \
        if (r_ != 0) return fail(MOCAP_E_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));             \
    } while (0)

extern "C" {

int mocap_abi_version(void) { return MOCAP_ABI_VERSION; }
const char* mocap_last_error(void) { return g_err.c_str(); }

int mocap_ctx_create(int device_id, int width, int height, int n_slots, mocap_ctx_t* out)
{
    if (!out || width < 1 || height < 1 || width > 32767 || height > 32767 || n_slots < 1 || n_slots > 64)
        return fail(MOCAP_E_INVALID, "mocap_ctx_create: bad geometry %dx%d slots=%d", width, height, n_slots);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(MOCAP_E_HIP, "mocap_ctx_create: no HIP device %d (%d visible)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    mocap_ctx* c = new mocap_ctx();
    c->device = device_id; c->W = width; c->H = height; c->n_slots = n_slots; c->wpr = (width + 31) / 32;
    c->prm = mocap_blob_params{5, 5, 255 * 0.85, 500.0, 0.5};
    c->maps = nullptr; c->map4 = nullptr; c->srcbox = nullptr; c->rowbox = nullptr; c->reach = nullptr; c->cflags = nullptr; c->map_flags = nullptr;
    c->mask = nullptr; c->mask_images = 0; c->mask_dirty = false; c->cells = nullptr; c->cells_images = 0; c->last_images = 0;
    c->hotmap = nullptr; c->tile_rows = nullptr; c->tile_rows_flip = 0; c->tile_rows_hold[0] = c->tile_rows_hold[1] = 0; c->cur_box = nullptr; c->items = nullptr; c->n_items = nullptr; c->cap_items = 0; c->wide_tiles = nullptr; c->cap_wide = 0;
    c->cells_ext = nullptr; c->cur_box_ext = nullptr; c->cells_ext_images = 0; c->cwork = nullptr; c->cwork_images = 0; c->walk_list = nullptr; c->link_list = nullptr; c->walk_count = nullptr;
    c->cams = nullptr; c->n_cam = 0; c->n_F = 0; c->scratch = nullptr; c->scratch_elems = 0; c->profiling = false;
    c->ba_obj = nullptr; c->ba_obj_elems = 0; c->ba_pinned = nullptr; c->ba_pinned_bytes = 0;
    c->comm.reset();
    c->side = nullptr; c->ev_fork = nullptr; c->ev_join = nullptr;
    c->tune = tuning_from_env();
    c->base_sel = c->tune.base_sel; c->probe_age = 0; c->probe_pending = false; c->probe_images = 0; c->hot_dense = false; c->probe_dev = nullptr; c->probe_host = nullptr; c->probe_ev = nullptr;
    {
        hipDeviceProp_t prop;
        c->box_grid = 2048; c->n_cu = 256;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) {
            c->box_grid = box_filter_blocks_per_cu() * prop.multiProcessorCount;
            c->n_cu = prop.multiProcessorCount;
            if (c->tune.box_blocks_per_cu >= 1) c->box_grid = c->tune.box_blocks_per_cu * prop.multiProcessorCount; // A/B switch
        }
    }
    c->slot_state.assign(n_slots, 0);
    c->slot_compact.assign(n_slots, 0);
    c->slot_wmax.assign(n_slots, 0);
    hipError_t e = hipMalloc(&c->map_flags, sizeof(uint32_t) * n_slots + 256);
    if (e == hipSuccess) e = hipMemset(c->map_flags, 0, sizeof(uint32_t) * n_slots + 256);
    if (e == hipSuccess) e = hipMalloc(&c->n_items, 1024); // item count + the 8 head words of the box kernel's runs
    if (e == hipSuccess) e = hipMemset(c->n_items, 0, 1024);
    if (e == hipSuccess) e = hipMalloc(&c->probe_dev, PROBE_BYTES);
    if (e == hipSuccess) e = hipHostMalloc(&c->probe_host, PROBE_BYTES);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->probe_ev, hipEventDisableTiming);
    // (the side stream of wide_fork is created on first use: every stream a process holds is dealt onto one of a few hardware
    // queues, and a stream nobody uses can end up sharing a queue with a batch's own stream)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->cams, sizeof(CameraTable));
    if (e == hipSuccess) e = hipMemset(c->cams, 0, sizeof(CameraTable));
    if (e != hipSuccess) {
        mocap_ctx_destroy(c);
        return fail(MOCAP_E_HIP, "mocap_ctx_create: %s", hipGetErrorString(e));
    }
    *out = c;
    return MOCAP_OK;
}

int mocap_ctx_destroy(mocap_ctx_t c)
{
    if (!c) return MOCAP_OK;
    (void)hipSetDevice(c->device);
    for (auto& v : c->ev) for (auto& p : v) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    (void)mocap_comm_destroy(c);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->probe_ev) { (void)hipEventSynchronize(c->probe_ev); (void)hipEventDestroy(c->probe_ev); }
    if (c->probe_dev) (void)hipFree(c->probe_dev);
    if (c->probe_host) (void)hipHostFree(c->probe_host);
    if (c->maps) (void)hipFree(c->maps);
    if (c->map4) (void)hipFree(c->map4);
    if (c->srcbox) (void)hipFree(c->srcbox);
    if (c->rowbox) (void)hipFree(c->rowbox);
    if (c->cur_box) (void)hipFree(c->cur_box);
    if (c->cur_box_ext) (void)hipFree(c->cur_box_ext);
    if (c->items) (void)hipFree(c->items);
    if (c->wide_tiles) (void)hipFree(c->wide_tiles);
    if (c->n_items) (void)hipFree(c->n_items);
    if (c->reach) (void)hipFree(c->reach);
    if (c->cflags) (void)hipFree(c->cflags);
    if (c->cells_ext) (void)hipFree(c->cells_ext);
    if (c->map_flags) (void)hipFree(c->map_flags);
    if (c->mask) (void)hipFree(c->mask);
    if (c->cells) (void)hipFree(c->cells);
    if (c->tile_rows) (void)hipFree(c->tile_rows);
    if (c->hotmap) (void)hipFree(c->hotmap);
    if (c->cwork) (void)hipFree(c->cwork);
    if (c->walk_list) (void)hipFree(c->walk_list);
    if (c->link_list) (void)hipFree(c->link_list);
    if (c->walk_count) (void)hipFree(c->walk_count);
    if (c->cams) (void)hipFree(c->cams);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->ba_obj) (void)hipFree(c->ba_obj);
    if (c->ba_pinned) (void)hipHostFree(c->ba_pinned);
    delete c;
    return MOCAP_OK;
}

int mocap_sync(mocap_ctx_t c, void* stream)
{
    if (!c) return fail(MOCAP_E_INVALID, "null context");
    if (set_device(c)) return MOCAP_E_HIP;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MOCAP_OK;
}

int mocap_set_blob_params(mocap_ctx_t c, const mocap_blob_params* p)
{
    if (!c || !p) return fail(MOCAP_E_INVALID, "null argument");
    if (p->ksize != 5 || p->median != 5)
        return fail(MOCAP_E_UNSUPPORTED, "only the reference's 5x5 blur and 5x5 median are implemented (got %d, %d)", p->ksize, p->median);
    if (!(p->thresh == p->thresh)) return fail(MOCAP_E_INVALID, "thresh is NaN");
    c->prm = *p;
    return MOCAP_OK;
}

int mocap_set_tuning(mocap_ctx_t c, const char* name, int value)
{
    if (!c || !name) return fail(MOCAP_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!strcmp(name, "rows") || !strcmp(name, "box_blocks_per_cu") || !strcmp(name, "base_sel"))
        return fail(MOCAP_E_STATE, "tuning '%s' shapes the context's buffers: set MOCAP_%s in the environment before mocap_ctx_create", name, name);
    if (!tune_set(c->tune, name, value)) return fail(MOCAP_E_INVALID, "unknown tuning name '%s'", name);
    return MOCAP_OK;
}

int mocap_set_undistort(mocap_ctx_t c, int slot, const double K[9], const double dist[5], int* identity_out)
{
    if (!c || !K || !dist) return fail(MOCAP_E_INVALID, "null argument");
    if (slot < 0 || slot >= c->n_slots) return fail(MOCAP_E_INVALID, "slot %d out of range", slot);
    if (set_device(c)) return MOCAP_E_HIP;
    std::lock_guard<std::mutex> lk(c->mu);
    size_t per = (size_t)c->H * c->W;
    if (!c->maps) HIP_TRY(hipMalloc(&c->maps, sizeof(uint32_t) * per * c->n_slots * 2));
    if (!c->map4) HIP_TRY(hipMalloc(&c->map4, sizeof(uint32_t) * (per * c->n_slots + 4))); // + 4: a quad load at the last pixel stays inside
    const int ncx_ = (c->W + 7) / 8, ncy_ = (c->H + 7) / 8;
    if (!c->srcbox) HIP_TRY(hipMalloc(&c->srcbox, sizeof(ushort4) * (size_t)ncx_ * ncy_ * c->n_slots));
    const int n_strips_ = tiling(c).n_strips;
    if (!c->rowbox) HIP_TRY(hipMalloc(&c->rowbox, sizeof(ushort4) * (size_t)c->H * n_strips_ * c->n_slots));
    MapArgs m;
    memcpy(m.K, K, sizeof(m.K));
    memcpy(m.dist, dist, sizeof(m.dist));
    m.H = c->H; m.W = c->W;
    m.map = c->maps + per * slot;
    m.mapw = c->maps + per * (c->n_slots + slot);
    m.map4 = c->map4 + per * slot;
    m.flags = c->map_flags + slot;
    HIP_TRY(hipMemset(m.flags, 0, sizeof(uint32_t)));
    launch_undistort_map(m, 0);
    HIP_TRY(hipGetLastError());
    uint32_t flags = 0;
    HIP_TRY(hipMemcpy(&flags, m.flags, sizeof(flags), hipMemcpyDeviceToHost));
    c->slot_state[slot] = (flags & 1u) ? 2 : 1;
    c->slot_compact[slot] = (flags & 2u) ? 0 : 1;
    c->slot_wmax[slot] = c->slot_state[slot] == 1 ? 1024u : 0u; // identity: every source pixel feeds exactly one output pixel
    launch_srcbox(m.map4, c->srcbox + (size_t)ncx_ * ncy_ * slot, c->H, c->W, 0);
    HIP_TRY(hipGetLastError());
    launch_rowbox(m.map4, c->rowbox + (size_t)c->H * n_strips_ * slot, c->H, c->W, n_strips_, 0);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> edge((size_t)ncx_ * ncy_, 0); // source cells read by windows that the image border cuts: bit 0 one axis, bit 1 both
    std::vector<int> reach32((size_t)ncx_ * ncy_ * 4);  // per source cell: x0, x1, y0, y1 of the output pixels that read it
    for (size_t i = 0; i < reach32.size(); i += 2) { reach32[i] = 0x7fffffff; reach32[i + 1] = -0x7fffffff - 1; }
    if (c->slot_state[slot] == 2) {
        // statistics for the dark-tile early-out (see blob_filter.hip): total weight per source pixel, tap extents
        uint32_t* tmp = nullptr;
        const size_t edge_words = edge.size();
        HIP_TRY(hipMalloc(&tmp, sizeof(uint32_t) * (per + 4 + edge_words + reach32.size())));
        hipError_t e2 = hipMemset(tmp, 0, sizeof(uint32_t) * (per + 4 + edge_words));
        int* reach_dev = (int*)(tmp + per + 4 + edge_words);
        if (e2 == hipSuccess) e2 = hipMemcpy(reach_dev, reach32.data(), sizeof(int) * reach32.size(), hipMemcpyHostToDevice);
        uint32_t st3[3] = {0, 0, 0};
        if (e2 == hipSuccess) {
            StatArgs sg{m.map, m.mapw, tmp, tmp + per, c->H, c->W, tmp + per + 4, reach_dev};
            launch_remap_stats(sg, 0);
            e2 = hipGetLastError();
            if (e2 == hipSuccess) e2 = hipMemcpy(st3, tmp + per, sizeof(st3), hipMemcpyDeviceToHost);
            if (e2 == hipSuccess) e2 = hipMemcpy(edge.data(), tmp + per + 4, sizeof(uint32_t) * edge.size(), hipMemcpyDeviceToHost);
            if (e2 == hipSuccess) e2 = hipMemcpy(reach32.data(), reach_dev, sizeof(int) * reach32.size(), hipMemcpyDeviceToHost);
        }
        (void)hipFree(tmp);
        if (e2 != hipSuccess) return fail(MOCAP_E_HIP, "undistort statistics: %s", hipGetErrorString(e2));
        if (st3[1] <= 9 && st3[2] <= 9 && c->W >= 8) c->slot_wmax[slot] = st3[0];
    }
    {   // Dark-tile early-out tables per 8x8 source cell: the reach (which output pixels read the cell: from the map itself
        // for a remapped camera, the cell's own pixels for the identity) and the border-cut window flags.
        const int H = c->H, W = c->W, ncx = ncx_, ncy = ncy_;
        std::vector<uint2> reach((size_t)ncx * ncy);
        std::vector<uint8_t> cflags((size_t)ncx * ncy);
        for (int cr = 0; cr < ncy; cr++)
            for (int cx = 0; cx < ncx; cx++) {
                const size_t i = (size_t)cr * ncx + cx;
                int x0, x1, y0, y1;
                if (c->slot_state[slot] == 1) {
                    x0 = 8 * cx; x1 = 8 * cx + 7 < W - 1 ? 8 * cx + 7 : W - 1; y0 = 8 * cr; y1 = 8 * cr + 7 < H - 1 ? 8 * cr + 7 : H - 1;
                    // identity: the cut windows lie within 4 pixels of the border
                    const bool xc = 8 * cx < 4 || 8 * cx + 7 >= W - 4, yc = 8 * cr < 4 || 8 * cr + 7 >= H - 4;
                    edge[i] = (xc && yc) ? 2u : (xc || yc) ? 1u : 0u;
                } else {
                    x0 = reach32[4 * i]; x1 = reach32[4 * i + 1]; y0 = reach32[4 * i + 2]; y1 = reach32[4 * i + 3];
                }
                if (x0 > x1 || y0 > y1) reach[i] = make_uint2(1u, 0u); // read by nothing: x0 = 1 > x1 = 0
                else reach[i] = make_uint2((uint32_t)x0 | ((uint32_t)x1 << 16), (uint32_t)y0 | ((uint32_t)y1 << 16));
                cflags[i] = (edge[i] & 2u) ? 2 : (edge[i] & 1u) ? 1 : 0;
            }
        if (!c->reach) HIP_TRY(hipMalloc(&c->reach, sizeof(uint2) * reach.size() * c->n_slots));
        HIP_TRY(hipMemcpy(c->reach + reach.size() * slot, reach.data(), sizeof(uint2) * reach.size(), hipMemcpyHostToDevice));
        if (!c->cflags) HIP_TRY(hipMalloc(&c->cflags, cflags.size() * c->n_slots));
        HIP_TRY(hipMemcpy(c->cflags + cflags.size() * slot, cflags.data(), cflags.size(), hipMemcpyHostToDevice));
    }
    if (identity_out) *identity_out = c->slot_state[slot] == 1;
    return MOCAP_OK;
}

int mocap_undistort_info(mocap_ctx_t c, int slot, mocap_undistort_info_t* out)
{
    if (!c || !out) return fail(MOCAP_E_INVALID, "null argument");
    if (slot < 0 || slot >= c->n_slots) return fail(MOCAP_E_INVALID, "slot %d out of range", slot);
    if (c->slot_state[slot] == 0) return fail(MOCAP_E_STATE, "mocap_set_undistort was not called for slot %d", slot);
    out->identity = c->slot_state[slot] == 1;
    out->compact_table = c->slot_compact[slot] != 0 && c->W >= 8;
    out->early_out_provable = c->slot_wmax[slot] != 0 && c->W >= 8;
    out->max_source_weight = (int32_t)c->slot_wmax[slot];
    // the sparse path (streaming scan + box kernel on the marked tiles) needs both; otherwise every tile of every image goes
    // through the dense row pipeline (same results, ~7x the time on a dark IR scene: DESIGN.md 4.1)
    out->sparse_path = out->compact_table && out->early_out_provable && c->tune.skip_dark && !c->tune.general_filter;
    return MOCAP_OK;
}

int mocap_set_cameras(mocap_ctx_t c, int n, const double* K, const double* dist, const double* R, const double* t)
{
    if (!c || !K || !dist || !R || !t) return fail(MOCAP_E_INVALID, "null argument");
    if (n < 1 || n > 32) return fail(MOCAP_E_INVALID, "camera count %d not in 1..32", n);
    if (set_device(c)) return MOCAP_E_HIP;
    HIP_TRY(hipMemcpy(&c->cams->K[0][0], K, sizeof(double) * 9 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&c->cams->dist[0][0], dist, sizeof(double) * 5 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&c->cams->R[0][0], R, sizeof(double) * 9 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(&c->cams->t[0][0], t, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
    c->n_cam = n;
    return MOCAP_OK;
}

int mocap_set_fundamentals(mocap_ctx_t c, int n, const double* F)
{
    if (!c || (n > 0 && !F)) return fail(MOCAP_E_INVALID, "null argument");
    if (n < 0 || n > 31) return fail(MOCAP_E_INVALID, "fundamental matrix count %d not in 0..31", n);
    if (set_device(c)) return MOCAP_E_HIP;
    if (n) HIP_TRY(hipMemcpy(&c->cams->F[0][0], F, sizeof(double) * 9 * n, hipMemcpyHostToDevice));
    c->n_F = n;
    return MOCAP_OK;
}

// ---- profiling -------------------------------------------------------------------------------------------------
static void prof_begin(mocap_ctx* c, int which, hipStream_t s, EvPair& p, bool& on)
{
    on = c->profiling;
    if (!on) return;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(p.a, s);
    (void)which;
}
static void prof_end(mocap_ctx* c, int which, hipStream_t s, EvPair& p, bool on)
{
    if (!on) return;
    (void)hipEventRecord(p.b, s);
    std::lock_guard<std::mutex> lk(c->mu);
    c->ev[which].push_back(p);
}

int mocap_profile_enable(mocap_ctx_t c, int on)
{
    if (!c) return fail(MOCAP_E_INVALID, "null context");
    c->profiling = on != 0;
    return MOCAP_OK;
}

int mocap_profile_read(mocap_ctx_t c, double ms[5], int cnt[5])
{
    if (!c || !ms || !cnt) return fail(MOCAP_E_INVALID, "null argument");
    if (set_device(c)) return MOCAP_E_HIP;
    for (int w = 0; w < 5; w++) { ms[w] = 0; cnt[w] = 0; }
    std::lock_guard<std::mutex> lk(c->mu);
    for (int w = 0; w < 5; w++) {
        for (auto& p : c->ev[w]) {
            HIP_TRY(hipEventSynchronize(p.b));
            float f = 0;
            HIP_TRY(hipEventElapsedTime(&f, p.a, p.b));
            ms[w] += f; cnt[w]++;
            (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b);
        }
        c->ev[w].clear();
    }
    return MOCAP_OK;
}

int mocap_tile_stats(mocap_ctx_t c, uint64_t* tiles, uint64_t* skipped)
{
    if (!c) return fail(MOCAP_E_INVALID, "null context");
    if (set_device(c)) return MOCAP_E_HIP;
    uint64_t total = 0, full = 0;
    if (c->cells && c->last_images > 0) {
        HIP_TRY(hipDeviceSynchronize());
        Tiling t = tiling(c);
        size_t per = cells_per_image(c), n = per * (size_t)c->last_images;
        std::vector<uint32_t> w(n);
        HIP_TRY(hipMemcpy(w.data(), c->cells, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        int valid_chunks = (c->H + t.rows - 1) / t.rows; // chunks that start inside the image
        total = (uint64_t)c->last_images * valid_chunks * t.n_strips;
        for (size_t i = 0; i < n; i++) full += w[i] >> 31;
    }
    if (tiles) *tiles = total;
    if (skipped) *skipped = total - full;
    return MOCAP_OK;
}

// ---- blob stage ------------------------------------------------------------------------------------------------
static int check_frames(mocap_ctx* c, const void* frames, int n_images, int cam_mod, int slot_base, size_t image_stride, int pitch)
{
    if (!c || !frames) return fail(MOCAP_E_INVALID, "null argument");
    if (n_images < 1) return fail(MOCAP_E_INVALID, "n_images = %d", n_images);
    if (cam_mod < 1 || slot_base < 0 || slot_base + cam_mod > c->n_slots)
        return fail(MOCAP_E_INVALID, "slots %d..%d not in 0..%d", slot_base, slot_base + cam_mod - 1, c->n_slots - 1);
    if (pitch < c->W) return fail(MOCAP_E_INVALID, "pitch %d < width %d", pitch, c->W);
    if (n_images > 1 && image_stride < (size_t)pitch * (c->H - 1) + c->W) return fail(MOCAP_E_INVALID, "image_stride too small");
    for (int s = slot_base; s < slot_base + cam_mod; s++)
        if (c->slot_state[s] == 0) return fail(MOCAP_E_STATE, "mocap_set_undistort was not called for slot %d", s);
    return 0;
}


// Excess base of the scan (BrightArgs::base): pixels count with max(0, p - c).  Exact for any c below the threshold; a
// higher c ignores brighter backgrounds, a lower c lets a cell hold more bright pixels before it is "hot" (tighter boxes
// around the markers).  Two candidates derived from the threshold -- for the reference's 216.75: 63 (tight; a dark IR
// frame) and 150 (backgrounds up to ~150 cost nothing) -- between which the context switches by itself: on its first batch
// and every 32nd one after it the scan also counts the cells that are hot under the other base (on every 16th image); the
// tight one is used whenever it does not leave noticeably more hot cells.  A context starts with the tolerant base (a
// bright scene filtered with the tight one would cost a dense pass).  MOCAP_EXCESS_BASE=c pins the base (A/B switch).
static int excess_base(int thr_mul, int sel)
{
    int c = sel ? thr_mul - 67 : thr_mul - 154;
    if (c > thr_mul - 1) c = thr_mul - 1;
    if (c > 254) c = 254;
    return c < 0 ? 0 : c;
}

// bayer != nullptr: `frames` (= bayer->dst) does not exist yet -- the Bayer -> gray pass that writes it runs first, fused
// with the streaming scan where the geometry allows (it has the gray bytes in registers anyway)
static int run_filter(mocap_ctx* c, const void* frames, int n_images, int cam_mod, int slot_base, size_t image_stride,
                      int pitch, uint32_t* mask, uint32_t* cells, hipStream_t s, const BayerArgs* bayer = nullptr)
{
    c->walk_count_zeroed = false;
    double ft = floor(c->prm.thresh);
    const int thr_mul = ft < -1.0 ? 0 : (ft > 255.0 ? 256 : (int)ft + 1);
    Tiling tl = tiling(c);
    const bool own_mask = mask == c->mask; // the context's mask keeps "zero outside the recorded regions" from batch to batch
    bool remap = false, compact = c->W >= 8 && cam_mod <= 64;
    uint64_t remap_bits = 0;
    for (int sl = slot_base; sl < slot_base + cam_mod; sl++) {
        if (c->slot_state[sl] == 2) { remap = true; if (sl - slot_base < 64) remap_bits |= 1ull << (sl - slot_base); }
        if (c->slot_state[sl] == 2 && !c->slot_compact[sl]) compact = false;
    }
    if (c->tune.general_filter) compact = false; // test switch: the general kernel
    // the row pipeline's staged form (compact table, source pixels through LDS) serves the dense path and the wide tiles alike
    bool rows_staged = remap && c->tune.rows_staged && c->map4 && c->rowbox && (c->W & 15) == 0 && c->H >= 2; // (16-byte staging units)
    for (int sl = slot_base; sl < slot_base + cam_mod; sl++)
        if (!c->slot_compact[sl]) rows_staged = false;
    const int rows_dw = c->tune.rows_stage_dw < 0 || c->tune.rows_stage_dw > rows_stage_dwords() ? rows_stage_dwords() : c->tune.rows_stage_dw;
    if (cells == c->cells) c->last_images = n_images;
    EvPair p; bool on;
    // dark-tile early-out: largest doubled excess sum 2E (E = sum of max(0, p - base)) per 16x16 block that still proves an
    // all-zero mask:   2E * Wmax < 1024 * taps_min * (2 * thr_mul - 2 * base - 1)     (derivation: blob_filter.hip)
    int fixed_base = c->tune.excess_base;
    if (fixed_base >= 0) { if (fixed_base > thr_mul - 1) fixed_base = thr_mul - 1; if (fixed_base > 254) fixed_base = 254; if (fixed_base < 0) fixed_base = 0; }
    if (c->probe_pending && hipEventQuery(c->probe_ev) == hipSuccess) { // the last probe's counts have arrived
        unsigned long long n_cur = 0, n_alt = 0;
        for (int i = 0; i < 128; i++) { n_cur += c->probe_host[PROBE_STRIDE * i]; n_alt += c->probe_host[PROBE_STRIDE * i + 1]; }
        if (c->tune.probe_debug) fprintf(stderr, "[probe] base %d: %llu hot cells, alternative %d: %llu\n", excess_base(thr_mul, c->base_sel), n_cur, excess_base(thr_mul, c->base_sel ^ 1), n_alt);
        // The tight base (sel 0) leaves tighter boxes around the markers for the same number of hot cells (measured: 33k against
        // 45k marked tiles per 3072 images of the benchmark scene), so it is preferred unless the background makes its hot cells
        // explode: use it iff it leaves at most 1.25x the hot cells of the tolerant base.
        const unsigned long long n_lo = c->base_sel == 0 ? n_cur : n_alt, n_hi = c->base_sel == 0 ? n_alt : n_cur;
        c->base_sel = (n_lo * 4 <= n_hi * 5) ? 0 : 1;
        // Who marks the tiles (scan_hotmap = 1: whichever is cheaper).  A hot cell costs the scan two dependent round trips behind
        // its loads; the hot map moves them into mark_tiles_kernel, which costs ~0.04 ms per 3072 images whatever the scene holds.
        // Measured (profiles/history/r4_run7_scan_wide_serial.log): scan + mark + settle 1.03 against 1.00 ms at 8 markers per frame
        // (~190 hot cells per image), 1.07 against 1.21 at 32 (~750): the map pays above a few hundred hot cells per image.
        const unsigned long long n_now = c->base_sel == 0 ? n_lo : n_hi;
        c->hot_dense = c->probe_images > 0 && n_now > 400ull * (unsigned long long)c->probe_images;
        c->probe_pending = false;
    }
    const int base = fixed_base >= 0 ? fixed_base : excess_base(thr_mul, c->base_sel);
    const int base_alt = excess_base(thr_mul, c->base_sel ^ 1);
    int allow = -1, allow_cut1 = -1, allow_cut2 = -1, allow_alt = -1;
    {
        long long wmax = 0;
        bool ok = true;
        for (int sl = slot_base; sl < slot_base + cam_mod; sl++) {
            if (c->slot_wmax[sl] == 0) ok = false;
            wmax = c->slot_wmax[sl] > wmax ? c->slot_wmax[sl] : wmax;
        }
        auto t5 = [](int n) { return (n - 1 < 2 ? n - 1 : 2) + 1; }; // taps of a window at the border, per axis
        auto t5full = [](int n) { return n < 5 ? n : 5; };
        const long long per_tap = 1024LL * (2LL * thr_mul - 2LL * base - 1), per_tap_alt = 1024LL * (2LL * thr_mul - 2LL * base_alt - 1);
        if (!c->tile_rows || !c->reach || !c->cflags) ok = false;
        if (ok && wmax > 0 && per_tap > 0) {
            allow = (int)((per_tap * t5full(c->W) * t5full(c->H) - 1) / wmax); // windows with all their taps
            const long long taps1 = t5(c->W) * t5full(c->H) < t5full(c->W) * t5(c->H) ? t5(c->W) * t5full(c->H) : t5full(c->W) * t5(c->H);
            allow_cut1 = (int)((per_tap * taps1 - 1) / wmax);                  // smallest window cut in one axis
            allow_cut2 = (int)((per_tap * t5(c->W) * t5(c->H) - 1) / wmax);    // smallest window cut in both
            if (per_tap_alt > 0) allow_alt = (int)((per_tap_alt * t5full(c->W) * t5full(c->H) - 1) / wmax);
        }
        if (!c->tune.skip_dark) allow = -1;
    }
    // every tile has to be filtered anyway: the dense kernel's sliding row pipeline does that with less work per pixel
    // than the box kernel (MOCAP_DENSE_BOXES=1: the box kernel on whole tiles, a test switch)
    if (allow < 0 && !c->tune.dense_boxes) compact = false;
    if (!compact) {
        // general dense kernel (tiny images, tables beyond the compact format): every tile, every mask byte
        FilterArgs a;
        a.src = (const uint8_t*)frames; a.image_stride = image_stride; a.pitch = pitch; a.H = c->H; a.W = c->W;
        a.mask = mask; a.words_per_row = c->wpr; a.cam_mod = cam_mod; a.cells = cells;
        a.map = c->maps ? c->maps + (size_t)slot_base * c->H * c->W : nullptr;
        a.mapw = c->maps ? c->maps + (size_t)(c->n_slots + slot_base) * c->H * c->W : nullptr;
        a.n_images = n_images; a.n_steps = (n_images + cam_mod - 1) / cam_mod;
        a.thr_mul = thr_mul;
        a.n_strips = tl.n_strips; a.rows_per_chunk = tl.rows; a.n_cgroups = tl.n_cgroups;
        a.pipelined = c->W >= 4 && (c->W & 3) == 0 && c->H >= 2;
        if (!c->tune.remap_pipeline) a.pipelined = 0; // test switch: the per-pixel gather
        a.staged = rows_staged; a.stage_dw = rows_dw;
        a.map4 = c->map4 ? c->map4 + (size_t)slot_base * c->H * c->W : nullptr;
        a.rowbox = c->rowbox ? c->rowbox + (size_t)slot_base * c->H * tl.n_strips : nullptr;
        if (bayer) { launch_bayer_gray(*bayer, s); HIP_TRY(hipGetLastError()); }
        if (own_mask) c->mask_dirty = true;
        prof_begin(c, 0, s, p, on);
        launch_filter_mask(a, remap, s);
        prof_end(c, 0, s, p, on);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (own_mask && c->mask_dirty) { // the general kernel wrote the whole mask last time: back to "zero outside the regions"
        HIP_TRY(hipMemsetAsync(c->mask, 0, sizeof(uint32_t) * c->mask_images * c->H * c->wpr, s));
        std::vector<uint32_t> init(c->mask_images * cells_per_image(c) * 4);
        for (size_t i = 0; i < init.size(); i += 4) { init[i] = 1u; init[i + 1] = 1u; init[i + 2] = 1u; init[i + 3] = 1u; }
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(c->cur_box, init.data(), sizeof(uint32_t) * init.size(), hipMemcpyHostToDevice));
        c->mask_dirty = false;
    }
    BoxArgs a;
    a.src = (const uint8_t*)frames; a.image_stride = image_stride; a.pitch = pitch; a.H = c->H; a.W = c->W;
    a.mask = mask; a.words_per_row = c->wpr; a.cells = cells;
    a.map4 = c->map4 ? c->map4 + (size_t)slot_base * c->H * c->W : nullptr;
    a.srcbox = c->srcbox ? c->srcbox + (size_t)slot_base * source_cells(c) : nullptr;
    a.remap_bits = remap_bits;
    a.cam_mod = cam_mod; a.n_images = n_images; a.n_steps = (n_images + cam_mod - 1) / cam_mod;
    a.thr_mul = thr_mul;
    a.rows_per_chunk = tl.rows; a.n_strips = tl.n_strips; a.n_chunks = tl.n_cgroups * 4;
    const size_t tr_words = c->mask_images * cells_per_image(c) * 4;
    uint32_t* const tr_cur = c->tile_rows + (c->tile_rows_flip ? tr_words : 0);
    a.tile_rows = tr_cur;
    a.tile_rows_next = c->tile_rows + (c->tile_rows_flip ? 0 : tr_words);
    a.n_clear = c->tile_rows_hold[c->tile_rows_flip ^ 1];
    a.cluster = c->tune.cluster;
    a.cur_box = own_mask ? c->cur_box : c->cur_box_ext;
    a.items = c->items; a.n_items = c->n_items; a.cap_items = c->cap_items;
    a.dense = allow < 0;
    // Boxes wider than this many patch quads go through the sliding row pipeline instead (whole tile width, the box's rows):
    // the box kernel's cost grows with the patch area (~30 cycles per quad-row), the row pipeline's with the rows only
    // (~850 cycles per row with the gather).  Measured optimum on the benchmark scenes (8 and 32 markers, both lens models):
    // 38-46 quads; without the routing the 32-marker scene's filter takes 1.77 ms instead of 1.25, the 8-marker scene's
    // 0.53 instead of 0.49.  MOCAP_WIDE_QUADS="remap,identity" overrides (A/B switch; 1000 = never).
    a.wide_tiles = c->wide_tiles; a.cap_wide = c->cap_wide; a.wide_quads_remap = c->tune.wide_quads_remap; a.wide_quads_identity = c->tune.wide_quads_identity;
    a.wide_bands = c->tune.wide_bands; // measured: 2 / 4 bands 0.50 / 0.55 ms against 0.475 (8 markers), 1.33 / 1.58 against 1.22 (32 markers): the kernel is work-bound
    if (c->W < 4) a.wide_tiles = nullptr;
    a.stage_bytes = c->tune.box_stage_bytes; // test switch
    a.prio = c->tune.box_prio;
    a.ext_mask = own_mask ? 0 : 1;
    if ((size_t)n_images * cells_per_image(c) * BOX_MAX_PARTS > (size_t)c->cap_items) return fail(MOCAP_E_STATE, "work list smaller than the batch");
    // the counter block (item counts, run heads) must be zero before settle: the scan's first workgroup does that on its way -- a fill
    // launch of its own is one more tiny kernel that waits for a place beside the other batches' kernels -- unless no plain scan runs
    const bool scan_zeroes = !a.dense && !bayer && c->tune.scan_blocks_per_cu == 0 && c->tune.scan_slices <= 1;
    if (!scan_zeroes) HIP_TRY(hipMemsetAsync(c->n_items, 0, 1024, s));
    a.zero8 = own_mask ? c->walk_count : nullptr; // (the contour stage of this batch follows on the same stream; null before its first batch)
    c->walk_count_zeroed = a.zero8 != nullptr;
    BrightArgs mark_args{}; bool mark_after_scan = false;
    if (!a.dense) { // one streaming pass over the frames marks the tiles (and their boxes) that can hold set pixels
        // floor(i / ncx) = umulhi(i, ceil(2^32 / ncx)) is exact while i * ncx < 2^32
        uint64_t ncx64 = (uint64_t)((c->W + 7) / 8);
        const uint64_t ncells = ncx64 * (uint64_t)((c->H + 7) / 8);
        int wide = (c->W % 16 == 0) && (pitch % 16 == 0) && (image_stride % 16 == 0) && (((uintptr_t)frames & 15) == 0) && ncx64 >= 4;
        if (wide && ncells * (ncx64 / 2) >= (1ull << 32)) wide = 0;
        if (!c->tune.scan_wide) wide = 0; // A/B switch
        if (wide) ncx64 /= 2; // the wide kernel divides pair indices by the pairs per cell row
        const uint32_t ncx_magic = (ncx64 > 1 && ncells * ncx64 < (1ull << 32)) ? (uint32_t)(((1ull << 32) + ncx64 - 1) / ncx64) : 0u;
        BrightArgs b{(const uint8_t*)frames, image_stride, pitch, c->H, c->W, n_images, cam_mod, ncx_magic, wide, base, allow / 4, allow_cut1 / 4, allow_cut2 / 4,
                     c->reach + (size_t)slot_base * source_cells(c), c->cflags + (size_t)slot_base * source_cells(c),
                     tr_cur, tl.n_cgroups * 4, tl.n_strips, (uint32_t)(((1u << 23) + tl.rows - 1) / tl.rows),
                     mask, own_mask ? 0 : (size_t)n_images * c->H * c->wpr, ((uintptr_t)mask & 15) == 0, nullptr, base_alt, allow_alt / 4, 0};
        b.prio = c->tune.scan_prio; // A/B switch
        b.max_blocks = c->tune.scan_blocks_per_cu * c->n_cu; b.blocks_x = 0;
        b.zero_counters = scan_zeroes ? c->n_items : nullptr;
        b.block_ctr = c->n_items + 160; // (words 160..223 of the counter block zeroed above: one per slice)
        b.slices = c->tune.scan_slices; b.image0 = 0; b.slice_images = n_images;
        const bool probe = fixed_base < 0 && !c->probe_pending && allow_alt >= 0 && base_alt != base && !bayer &&
                           (c->probe_age == 0 || c->probe_age >= 32);
        if (probe) {
            HIP_TRY(hipMemsetAsync(c->probe_dev, 0, PROBE_BYTES, s));
            b.probe = c->probe_dev;
        }
        c->probe_age = probe ? 1 : c->probe_age + 1;
        const bool fused = bayer && own_mask && bayer_scan_fusable(*bayer);
        // the streaming scan leaves a hot map (two bits per cell, no table lookups or atomics behind its loads) that
        // mark_tiles_kernel turns into tile boxes; the fused Bayer pass marks the tiles itself (MOCAP_SCAN_HOTMAP=0: so does the scan)
        const bool two_step = !fused && c->hotmap && (c->tune.scan_hotmap == 2 || (c->tune.scan_hotmap == 1 && c->hot_dense));
        if (two_step) { b.hotmap = c->hotmap; b.hot_words = hot_map_words(c->H, c->W, wide); b.mark_grid = c->tune.mark_blocks_per_cu * c->n_cu; }
        ScanTurn* turn = c->tune.scan_serial && c->device >= 0 && c->device < 64 ? &g_scan_turn[c->device] : nullptr;
        std::unique_lock<std::mutex> turn_lock;
        if (turn) {
            turn_lock = std::unique_lock<std::mutex>(turn->mu);
            if (!turn->done) HIP_TRY(hipEventCreateWithFlags(&turn->done, hipEventDisableTiming));
            if (turn->have) HIP_TRY(hipStreamWaitEvent(s, turn->done, 0));
        }
        prof_begin(c, 3, s, p, on);
        if (fused) launch_bayer_gray_scan(*bayer, b, s);
        else {
            if (bayer) launch_bayer_gray(*bayer, s);
            launch_bright_cells(b, s);
        }
        prof_end(c, 3, s, p, on);
        HIP_TRY(hipGetLastError());
        if (turn) {
            HIP_TRY(hipEventRecord(turn->done, s));
            turn->have = true;
            turn_lock.unlock();
        }
        mark_args = b; mark_after_scan = two_step; // (launched with settle, inside its timer: both turn the scan's output into work lists)
        if (probe) {
            HIP_TRY(hipMemcpyAsync(c->probe_host, c->probe_dev, PROBE_BYTES, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(c->probe_ev, s));
            c->probe_pending = true;
            c->probe_images = (n_images + 15) / 16;
        }
    }
    else if (bayer) { // no early-out (not provable for this table, or MOCAP_SKIP_DARK=0): the plain gray pass
        launch_bayer_gray(*bayer, s);
        HIP_TRY(hipGetLastError());
    }
    prof_begin(c, 4, s, p, on);
    if (mark_after_scan) {
        launch_mark_tiles(mark_args, s);
        HIP_TRY(hipGetLastError());
    }
    launch_settle_tiles(a, s);
    prof_end(c, 4, s, p, on);
    HIP_TRY(hipGetLastError());
    if (!a.dense) { // only now: settle (queued) has emptied the array the next batch's scan will widen
        if (n_images > c->tile_rows_hold[c->tile_rows_flip]) c->tile_rows_hold[c->tile_rows_flip] = n_images;
        c->tile_rows_hold[c->tile_rows_flip ^ 1] = 0;
        c->tile_rows_flip ^= 1;
    }
    a.timing = nullptr;
    const bool box_timing = c->tune.box_timing != 0;
    if (box_timing) { // debugging aid: synchronous, prints the mean duration of the box kernel's phases
        HIP_TRY(hipMalloc(&a.timing, sizeof(uint64_t) * 6 * c->box_grid));
        HIP_TRY(hipMemsetAsync(a.timing, 0, sizeof(uint64_t) * 6 * c->box_grid, s));
    }
    prof_begin(c, 0, s, p, on);
    // MOCAP_WIDE_FORK=1: the wide-tile kernel on a side stream beside the box kernel (fork / join by events).  Measured: the pair takes
    // 0.53 ms instead of 0.49 alone and the three-batch pipeline 310k instead of 319k frames/s, so it is off.
    const bool fork_wide = c->tune.wide_fork != 0;
    if (a.wide_tiles) { // the tiles with wide boxes: the row pipeline over their list, beside the box kernel (both only read what
                        // settle left and write disjoint tiles): forked onto the context's side stream, joined before the contours
        FilterArgs f;
        f.src = a.src; f.image_stride = image_stride; f.pitch = pitch; f.H = c->H; f.W = c->W;
        f.mask = mask; f.words_per_row = c->wpr; f.cam_mod = cam_mod; f.cells = cells;
        f.map = c->maps ? c->maps + (size_t)slot_base * c->H * c->W : nullptr;
        f.mapw = c->maps ? c->maps + (size_t)(c->n_slots + slot_base) * c->H * c->W : nullptr;
        f.n_images = n_images; f.n_steps = a.n_steps; f.thr_mul = thr_mul;
        f.n_strips = tl.n_strips; f.rows_per_chunk = tl.rows; f.n_cgroups = tl.n_cgroups;
        f.tiles = c->wide_tiles; f.n_tiles = c->n_items + 8; f.cap_tiles = c->cap_wide;
        f.pipelined = (c->W & 3) == 0 && c->H >= 2;
        f.staged = rows_staged; f.stage_dw = rows_dw; f.map4 = a.map4;
        f.rowbox = c->rowbox ? c->rowbox + (size_t)slot_base * c->H * tl.n_strips : nullptr;
        hipStream_t ws = s;
        if (fork_wide) {
            HIP_TRY(hipEventRecord(c->ev_fork, s));
            if (!c->side) HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
            HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
            ws = c->side;
        }
        launch_filter_tiles(f, remap, c->tune.wide_blocks_per_cu * c->n_cu, ws);
        HIP_TRY(hipGetLastError());
        if (fork_wide) HIP_TRY(hipEventRecord(c->ev_join, c->side));
    }
    launch_box_filter(a, c->box_grid, s);
    HIP_TRY(hipGetLastError());
    if (a.wide_tiles && fork_wide) HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
    prof_end(c, 0, s, p, on);
    if (box_timing) {
        std::vector<uint64_t> t((size_t)6 * c->box_grid);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(t.data(), a.timing, sizeof(uint64_t) * t.size(), hipMemcpyDeviceToHost));
        (void)hipFree(a.timing);
        double sum[6] = {0, 0, 0, 0, 0, 0}, mx = 0;
        for (int b = 0; b < c->box_grid; b++) {
            double tot = 0;
            for (int i = 0; i < 6; i++) { sum[i] += (double)t[6 * b + i]; if (i < 5) tot += (double)t[6 * b + i]; }
            mx = tot > mx ? tot : mx;
        }
        const double n = sum[5] > 0 ? sum[5] : 1;
        uint32_t cnt[16];
        HIP_TRY(hipMemcpy(cnt, c->n_items, sizeof(cnt), hipMemcpyDeviceToHost));
        fprintf(stderr, "[box] list: %u items, %u wide-tile entries\n", cnt[0], cnt[8]);
        fprintf(stderr, "[box] items %.0f (%.1f per wave) | cycles per item: header+wait %.0f, stage %.0f, patch %.0f, threshold %.0f, majority+next %.0f | busiest wave %.0f cycles\n",
                sum[5], sum[5] / c->box_grid, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, mx);
    }
    return 0;
}

// the tiles' output regions / scan boxes of the batch just filtered into the context's mask (settle_tiles_kernel), when the
// sparse path ran (the general dense kernel keeps none); MOCAP_CONTOUR_BOXES=0: whole strips (A/B switch, same results)
static const uint32_t* contour_boxes(mocap_ctx* c)
{
    return (c->mask_dirty || !c->tune.contour_boxes) ? nullptr : c->cur_box;
}

static int run_contours(mocap_ctx* c, const uint32_t* mask, const uint32_t* cells, const uint32_t* boxes, int n_images, int32_t* out_xy, long xy_stride,
                        int32_t* out_count, long count_stride, int max_blobs, mocap_contour* dbg, int32_t* dbg_count, int dbg_cap, hipStream_t s)
{
    ContourArgs a;
    a.mask = mask; a.words_per_row = c->wpr; a.H = c->H; a.W = c->W; a.n_images = n_images;
    a.out_xy = out_xy; a.out_count = out_count; a.max_blobs = max_blobs;
    a.xy_stride = xy_stride; a.count_stride = count_stride;
    a.min_area = c->prm.min_area; a.min_circ = c->prm.min_circ;
    a.dbg = (ContourRec*)dbg; a.dbg_count = dbg_count; a.dbg_cap = dbg_cap;
    long long ms = 4LL * c->H * c->W + 16;
    a.max_steps = ms > (1 << 22) ? (1 << 22) : (int)ms;
    Tiling tl = tiling(c);
    a.cells = cells; a.boxes = boxes; a.rows_per_chunk = tl.rows; a.n_chunks = tl.n_cgroups * 4; a.n_strips = tl.n_strips;
    if ((long long)a.n_chunks * a.n_strips * ((tl.rows + 7) / 8) > 65535 || c->wpr > 4096) a.cells = nullptr; // cell ids are 16-bit, first words 12-bit in the kernel: scan every row instead
    if ((size_t)n_images > c->cwork_images) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->cwork) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->cwork)); c->cwork = nullptr; c->cwork_images = 0; }
        if (c->walk_list) { HIP_TRY(hipFree(c->walk_list)); c->walk_list = nullptr; }
        if (c->link_list) { HIP_TRY(hipFree(c->link_list)); c->link_list = nullptr; }
        HIP_TRY(hipMalloc(&c->cwork, contour_work_bytes() * (size_t)n_images));
        HIP_TRY(hipMalloc(&c->walk_list, contour_walk_bytes() * (size_t)n_images));
        HIP_TRY(hipMalloc(&c->link_list, (contour_link_bytes() + sizeof(uint32_t)) * (size_t)n_images)); // + the wait list behind it
        if (!c->walk_count) HIP_TRY(hipMalloc(&c->walk_count, 256));
        c->cwork_images = n_images;
    }
    a.work = c->cwork;
    // The split form (candidates per image -> all walks of the batch, 64 to a wave -> tree per image) is the default;
    // MOCAP_CONTOURS_SPLIT=0 runs the one-kernel-per-image form (A/B switch; same results).
    const bool split = c->tune.contours_split != 0;
    a.walk_list = split ? c->walk_list : nullptr; a.link_list = c->link_list; a.walk_count = c->walk_count;
    a.follow_grid = c->n_cu * 4;  // 4 one-wave workgroups per CU (33 KB of LDS each): persistent, they refill their lanes from the list
    a.follow_grid2 = c->n_cu;     // the link walks are few
    a.image_grid = c->tune.contour_blocks_per_cu * c->n_cu;
    a.counters_zeroed = mask == c->mask && c->walk_count_zeroed; // (settle of this batch, same stream)
    c->walk_count_zeroed = false;
    // Links that only a walk can settle (nested rings, overlapping boxes): deferred to a second, packed follow pass + a second tree pass
    // -- two more launches per batch, nearly always empty on frames of separate markers, and beside another batch's scan an empty
    // launch costs up to 0.3 ms (profiles/history/r4_timeline_depth3.txt) -- or walked in place by the first tree pass (one wave per
    // link: 0.1-0.2 ms when a crowded batch holds a long one).  contour_defer = 1: deferred only once a probe found the scene crowded.
    a.defer_links = c->tune.contour_defer == 2 || (c->tune.contour_defer == 1 && c->hot_dense);
    a.wait_list = (uint32_t*)((uint8_t*)c->link_list + contour_link_bytes() * c->cwork_images);
    a.follow_list = 0; a.tree_pass = 0; a.follow_dbg = nullptr; a.follow_dbg_list = c->tune.follow_timing == 2 ? 1 : 0;
    if (c->tune.follow_timing && split) {
        HIP_TRY(hipMalloc(&a.follow_dbg, sizeof(uint64_t) * 8 * a.follow_grid));
        HIP_TRY(hipMemsetAsync(a.follow_dbg, 0, sizeof(uint64_t) * 8 * a.follow_grid, s));
    }
    a.prio = c->tune.contour_prio; // A/B switch (no effect measured)
    a.timing = nullptr;
    const bool phase_timing = c->tune.contour_timing != 0;
    if (phase_timing) { // debugging aid: synchronous, prints the mean duration of the kernel's phases
        HIP_TRY(hipMalloc(&a.timing, sizeof(uint64_t) * 8 * n_images));
        HIP_TRY(hipMemsetAsync(a.timing, 0, sizeof(uint64_t) * 8 * n_images, s));
    }
    EvPair p; bool on;
    prof_begin(c, 1, s, p, on);
    launch_contours(a, s);
    prof_end(c, 1, s, p, on);
    HIP_TRY(hipGetLastError());
    if (a.follow_dbg) {
        std::vector<uint64_t> t((size_t)8 * a.follow_grid);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(t.data(), a.follow_dbg, sizeof(uint64_t) * t.size(), hipMemcpyDeviceToHost));
        (void)hipFree(a.follow_dbg);
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx[3] = {0, 0, 0}, mxsteps = 0; int used = 0;
        for (int b = 0; b < a.follow_grid; b++) {
            if (t[8 * b + 5] == 0) continue;
            used++;
            for (int i = 0; i < 8; i++) sum[i] += (double)t[8 * b + i];
            for (int i = 0; i < 3; i++) mx[i] = (double)t[8 * b + i] > mx[i] ? (double)t[8 * b + i] : mx[i];
            mxsteps = (double)t[8 * b + 3] > mxsteps ? (double)t[8 * b + 3] : mxsteps;
        }
        const double u = used ? used : 1;
        fprintf(stderr, "[follow] %d of %d waves had work | per wave (mean / max us): store %.1f / %.1f, refill %.1f / %.1f, walk %.1f / %.1f | wave steps %.0f (max %.0f), "
                        "lanes alive per step %.1f, walks %.1f, refills %.1f | us per wave step %.3f\n",
                used, a.follow_grid, sum[0] / u / 100, mx[0] / 100, sum[1] / u / 100, mx[1] / 100, sum[2] / u / 100, mx[2] / 100, sum[3] / u, mxsteps,
                sum[3] > 0 ? sum[4] / sum[3] : 0.0, sum[5] / u, sum[6] / u, sum[3] > 0 ? sum[2] / 100 / sum[3] : 0.0);
    }
    if (phase_timing) {
        std::vector<uint64_t> t((size_t)8 * n_images);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(t.data(), a.timing, sizeof(uint64_t) * t.size(), hipMemcpyDeviceToHost));
        (void)hipFree(a.timing);
        double sum[4] = {0, 0, 0, 0}, mx = 0; uint64_t lo = ~0ull, hi = 0;
        for (int i = 0; i < n_images; i++) {
            for (int k = 0; k < 4; k++) sum[k] += (double)(t[8 * i + k + 1] - t[8 * i + k]);
            double tot = (double)(t[8 * i + 4] - t[8 * i]); mx = tot > mx ? tot : mx;
            lo = t[8 * i] < lo ? t[8 * i] : lo; hi = t[8 * i + 4] > hi ? t[8 * i + 4] : hi;
        }
        {
            int worst = 0; double wt = 0, sc = 0, ss = 0;
            for (int i = 0; i < n_images; i++) {
                double tot = (double)(t[8 * i + 4] - t[8 * i]);
                if (tot > wt) { wt = tot; worst = i; }
                sc += (double)t[8 * i + 5]; ss += (double)t[8 * i + 6];
            }
            fprintf(stderr, "[contours] slowest image %d: %.1f us, candidates %llu, longest border %llu steps, borders %llu | mean candidates %.1f, mean longest border %.1f steps\n",
                    worst, wt / 100, (unsigned long long)t[8 * worst + 5], (unsigned long long)t[8 * worst + 6], (unsigned long long)t[8 * worst + 7],
                    sc / n_images, ss / n_images);
        }
        fprintf(stderr, "[contours] mean us per block: candidates %.1f follow %.1f link %.1f order %.1f | slowest block %.1f | first start to last end %.1f\n",
                sum[0] / n_images / 100, sum[1] / n_images / 100, sum[2] / n_images / 100, sum[3] / n_images / 100, mx / 100, (double)(hi - lo) / 100);
    }
    return 0;
}

static int ensure_mask(mocap_ctx* c, int n_images)
{
    if ((size_t)n_images <= c->mask_images) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    if ((size_t)n_images <= c->mask_images) return 0;
    if (c->mask) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->mask)); c->mask = nullptr; c->mask_images = 0; c->last_images = 0; }
    size_t bytes = sizeof(uint32_t) * (size_t)n_images * c->H * c->wpr;
    HIP_TRY(hipMalloc(&c->mask, bytes));
    HIP_TRY(hipMemset(c->mask, 0, bytes));
    c->mask_images = n_images;
    if (c->cells) { HIP_TRY(hipFree(c->cells)); c->cells = nullptr; }
    size_t cbytes = sizeof(uint32_t) * (size_t)n_images * cells_per_image(c);
    HIP_TRY(hipMalloc(&c->cells, cbytes));
    HIP_TRY(hipMemset(c->cells, 0, cbytes));
    c->cells_images = n_images;
    if (c->tile_rows) { HIP_TRY(hipFree(c->tile_rows)); c->tile_rows = nullptr; }
    if (c->hotmap) { HIP_TRY(hipFree(c->hotmap)); c->hotmap = nullptr; }
    if (c->cur_box) { HIP_TRY(hipFree(c->cur_box)); c->cur_box = nullptr; }
    if (c->items) { HIP_TRY(hipFree(c->items)); c->items = nullptr; c->cap_items = 0; }
    if (c->wide_tiles) { HIP_TRY(hipFree(c->wide_tiles)); c->wide_tiles = nullptr; c->cap_wide = 0; }
    {   // every tile starts with the empty box (0xffffffff, 0) and an empty recorded region (x0 = 1 > x1 = 0)
        std::vector<uint32_t> init((size_t)n_images * cells_per_image(c) * 4);
        for (size_t i = 0; i < init.size(); i += 2) { init[i] = 0xffffffffu; init[i + 1] = 0u; }
        HIP_TRY(hipMalloc(&c->tile_rows, 2 * sizeof(uint32_t) * init.size()));
        HIP_TRY(hipMemcpy(c->tile_rows, init.data(), sizeof(uint32_t) * init.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->tile_rows + init.size(), init.data(), sizeof(uint32_t) * init.size(), hipMemcpyHostToDevice));
        c->tile_rows_flip = 0; c->tile_rows_hold[0] = c->tile_rows_hold[1] = 0;
        // the hot map, sized for either scan form
        const int hw0 = hot_map_words(c->H, c->W, 0), hw1 = hot_map_words(c->H, c->W, 1);
        HIP_TRY(hipMalloc(&c->hotmap, sizeof(uint32_t) * (size_t)n_images * (hw0 > hw1 ? hw0 : hw1)));
        HIP_TRY(hipMemset(c->hotmap, 0, sizeof(uint32_t) * (size_t)n_images * (hw0 > hw1 ? hw0 : hw1))); // all zeros between batches: the scan stores hot words only, mark_tiles_kernel clears them
        for (size_t i = 0; i < init.size(); i++) init[i] = 1u;
        HIP_TRY(hipMalloc(&c->cur_box, sizeof(uint32_t) * init.size()));
        HIP_TRY(hipMemcpy(c->cur_box, init.data(), sizeof(uint32_t) * init.size(), hipMemcpyHostToDevice));
        const size_t cap = (size_t)n_images * cells_per_image(c) * BOX_MAX_PARTS; // settle_tiles_kernel cuts a tile into at most that many items
        if (cap > 0xffffffffull) return fail(MOCAP_E_UNSUPPORTED, "batch too large for the work list");
        HIP_TRY(hipMalloc(&c->items, sizeof(BoxItem) * cap));
        c->cap_items = (uint32_t)cap;
        HIP_TRY(hipMalloc(&c->wide_tiles, sizeof(uint4) * 4 * (size_t)n_images * cells_per_image(c))); // up to 4 row bands per tile
        c->cap_wide = (uint32_t)(4 * (size_t)n_images * cells_per_image(c));
        c->mask_dirty = false;
    }
    return 0;
}

int mocap_filter_mask(mocap_ctx_t c, const void* frames, int n_images, int cam_mod, int slot_base, size_t image_stride,
                      int pitch, uint32_t* mask_dev, void* stream)
{
    int rc = check_frames(c, frames, n_images, cam_mod, slot_base, image_stride, pitch);
    if (rc) return rc;
    if (!mask_dev) return fail(MOCAP_E_INVALID, "null mask");
    if (set_device(c)) return MOCAP_E_HIP;
    if ((rc = ensure_mask(c, n_images))) return rc; // for the tile flags
    if ((size_t)n_images > c->cells_ext_images) { // occupancy words of a caller-owned mask: never mixed with the context's own
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->cells_ext) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->cells_ext)); c->cells_ext = nullptr; c->cells_ext_images = 0; }
        HIP_TRY(hipMalloc(&c->cells_ext, sizeof(uint32_t) * (size_t)n_images * cells_per_image(c)));
        if (c->cur_box_ext) { HIP_TRY(hipFree(c->cur_box_ext)); c->cur_box_ext = nullptr; }
        HIP_TRY(hipMalloc(&c->cur_box_ext, sizeof(uint32_t) * 4 * (size_t)n_images * cells_per_image(c)));
        c->cells_ext_images = n_images;
    }
    return run_filter(c, frames, n_images, cam_mod, slot_base, image_stride, pitch, mask_dev, c->cells_ext, (hipStream_t)stream);
}

int mocap_contours_from_mask(mocap_ctx_t c, const uint32_t* mask_dev, int n_images, int32_t* out_xy, long xy_stride,
                             int32_t* out_count, long count_stride, int max_blobs, mocap_contour* dbg, int32_t* dbg_count,
                             int dbg_cap, void* stream)
{
    if (!c || !mask_dev || !out_xy || !out_count) return fail(MOCAP_E_INVALID, "null argument");
    if (n_images < 1 || max_blobs < 1 || xy_stride < 2L * max_blobs || count_stride < 1)
        return fail(MOCAP_E_INVALID, "n_images=%d max_blobs=%d strides %ld %ld", n_images, max_blobs, xy_stride, count_stride);
    if ((dbg != nullptr) != (dbg_count != nullptr) || (dbg && dbg_cap < 1)) return fail(MOCAP_E_INVALID, "inconsistent debug buffers");
    if (set_device(c)) return MOCAP_E_HIP;
    return run_contours(c, mask_dev, nullptr, nullptr, n_images, out_xy, xy_stride, out_count, count_stride, max_blobs, dbg, dbg_count, dbg_cap,
                        (hipStream_t)stream);
}

int mocap_blob_centroids(mocap_ctx_t c, const void* frames, int n_images, int cam_mod, int slot_base, size_t image_stride,
                         int pitch, int32_t* out_xy, long xy_stride, int32_t* out_count, long count_stride, int max_blobs,
                         void* stream)
{
    int rc = check_frames(c, frames, n_images, cam_mod, slot_base, image_stride, pitch);
    if (rc) return rc;
    if (!out_xy || !out_count || max_blobs < 1 || xy_stride < 2L * max_blobs || count_stride < 1)
        return fail(MOCAP_E_INVALID, "bad output arguments");
    if (set_device(c)) return MOCAP_E_HIP;
    if ((rc = ensure_mask(c, n_images))) return rc;
    if ((rc = run_filter(c, frames, n_images, cam_mod, slot_base, image_stride, pitch, c->mask, c->cells, (hipStream_t)stream))) return rc;
    return run_contours(c, c->mask, c->cells, contour_boxes(c), n_images, out_xy, xy_stride, out_count, count_stride, max_blobs, nullptr, nullptr, 0,
                        (hipStream_t)stream);
}

static int bayer_args(BayerArgs& a, const void* bayer, void* gray, int n_images, int H, int W, long spitch, long dpitch,
                      size_t src_image_stride, size_t dst_image_stride, int pattern, int gray_shift);

int mocap_blob_centroids_bayer(mocap_ctx_t c, const void* bayer_frames, void* gray_frames, int n_images, int cam_mod, int slot_base,
                               size_t image_stride, int pitch, int pattern, int gray_shift, int32_t* out_xy, long xy_stride,
                               int32_t* out_count, long count_stride, int max_blobs, void* stream)
{
    int rc = check_frames(c, bayer_frames, n_images, cam_mod, slot_base, image_stride, pitch);
    if (rc) return rc;
    if (!out_xy || !out_count || max_blobs < 1 || xy_stride < 2L * max_blobs || count_stride < 1)
        return fail(MOCAP_E_INVALID, "bad output arguments");
    BayerArgs b;
    if ((rc = bayer_args(b, bayer_frames, gray_frames, n_images, c->H, c->W, pitch, pitch, image_stride, image_stride, pattern, gray_shift)))
        return rc;
    if (set_device(c)) return MOCAP_E_HIP;
    if ((rc = ensure_mask(c, n_images))) return rc;
    if ((rc = run_filter(c, gray_frames, n_images, cam_mod, slot_base, image_stride, pitch, c->mask, c->cells, (hipStream_t)stream, &b)))
        return rc;
    return run_contours(c, c->mask, c->cells, contour_boxes(c), n_images, out_xy, xy_stride, out_count, count_stride, max_blobs, nullptr, nullptr, 0,
                        (hipStream_t)stream);
}

int mocap_undistort_u8(mocap_ctx_t c, int slot, const void* src, void* dst, int spitch, int dpitch, void* stream)
{
    if (!c || !src || !dst) return fail(MOCAP_E_INVALID, "null argument");
    if (slot < 0 || slot >= c->n_slots || c->slot_state[slot] == 0) return fail(MOCAP_E_STATE, "undistort slot %d not set", slot);
    if (spitch < c->W || dpitch < c->W) return fail(MOCAP_E_INVALID, "pitch < width");
    if (set_device(c)) return MOCAP_E_HIP;
    launch_undistort((const uint8_t*)src, (uint8_t*)dst, c->H, c->W, spitch, dpitch, c->maps + (size_t)slot * c->H * c->W,
                     c->maps + (size_t)(c->n_slots + slot) * c->H * c->W, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

int mocap_image_filter_u8(mocap_ctx_t c, const void* src, void* dst, int spitch, int dpitch, int order, int slot, void* stream)
{
    if (!c || !src || !dst) return fail(MOCAP_E_INVALID, "null argument");
    if (order != 0 && order != 1) return fail(MOCAP_E_INVALID, "order must be 0 (image_filter_gpu) or 1 (image_filter_cpu)");
    if (spitch < c->W || dpitch < c->W) return fail(MOCAP_E_INVALID, "pitch < width");
    if (slot >= c->n_slots || (slot >= 0 && c->slot_state[slot] == 0)) return fail(MOCAP_E_STATE, "undistort slot %d not set", slot);
    if (set_device(c)) return MOCAP_E_HIP;
    hipStream_t s = (hipStream_t)stream;
    double ft = floor(c->prm.thresh);
    int ithresh = ft < -1.0 ? -1 : (ft > 255.0 ? 255 : (int)ft);
    if (order == 1) {
        const uint8_t* in = (const uint8_t*)src;
        int ip = spitch;
        if (slot >= 0 && c->slot_state[slot] == 2) { // undistort into dst, then filter in a second buffer
            return fail(MOCAP_E_UNSUPPORTED, "image_filter_cpu order with undistortion: call mocap_undistort_u8 first");
        }
        launch_median5(in, (uint8_t*)dst, c->H, c->W, ip, dpitch, ithresh, 1, s);
        HIP_TRY(hipGetLastError());
        return MOCAP_OK;
    }
    int rc = ensure_mask(c, 1);
    if (rc) return rc;
    // the general kernel with a one-image batch; slot < 0 = no undistortion
    FilterArgs a;
    a.src = (const uint8_t*)src; a.image_stride = 0; a.pitch = spitch; a.H = c->H; a.W = c->W;
    a.mask = c->mask; a.words_per_row = c->wpr; a.cam_mod = 1; a.n_images = 1; a.n_steps = 1;
    a.cells = c->cells;
    a.map = slot >= 0 ? c->maps + (size_t)slot * c->H * c->W : nullptr;
    a.mapw = slot >= 0 ? c->maps + (size_t)(c->n_slots + slot) * c->H * c->W : nullptr;
    a.thr_mul = ithresh + 1;
    Tiling tl = tiling(c);
    a.n_strips = tl.n_strips; a.rows_per_chunk = tl.rows; a.n_cgroups = tl.n_cgroups;
    a.pipelined = c->W >= 4 && (c->W & 3) == 0 && c->H >= 2;
    c->mask_dirty = true; c->last_images = 1;
    launch_filter_mask(a, slot >= 0 && c->slot_state[slot] == 2, s);
    HIP_TRY(hipGetLastError());
    launch_mask_expand(c->mask, c->wpr, (uint8_t*)dst, c->H, c->W, dpitch, s);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

int mocap_box_blur_u8(mocap_ctx_t c, const void* src, void* dst, int H, int W, int spitch, int dpitch, int ksize, void* stream)
{
    if (!c || !src || !dst) return fail(MOCAP_E_INVALID, "null argument");
    if (H < 1 || W < 1 || spitch < W || dpitch < W || ksize < 1 || ksize > 31) return fail(MOCAP_E_INVALID, "bad geometry");
    if (set_device(c)) return MOCAP_E_HIP;
    launch_box_blur((const uint8_t*)src, (uint8_t*)dst, H, W, spitch, dpitch, ksize, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

static int bayer_args(BayerArgs& a, const void* bayer, void* gray, int n_images, int H, int W, long spitch, long dpitch,
                      size_t src_image_stride, size_t dst_image_stride, int pattern, int gray_shift)
{
    if (!bayer || !gray) return fail(MOCAP_E_INVALID, "null argument");
    if (n_images < 1 || n_images > 65535 || H < 3 || W < 3 || spitch < W || dpitch < W)
        return fail(MOCAP_E_INVALID, "bad geometry: n=%d H=%d W=%d pitches %ld %ld (H, W >= 3)", n_images, H, W, spitch, dpitch);
    if (n_images > 1 && (src_image_stride < (size_t)spitch * (H - 1) + W || dst_image_stride < (size_t)dpitch * (H - 1) + W))
        return fail(MOCAP_E_INVALID, "image strides smaller than an image");
    if (pattern < 0 || pattern > 3 || (gray_shift != 14 && gray_shift != 15))
        return fail(MOCAP_E_INVALID, "pattern %d (0..3 = BG, GB, RG, GR) / gray_shift %d (14 or 15)", pattern, gray_shift);
    a = BayerArgs{};
    a.src = (const uint8_t*)bayer; a.dst = (uint8_t*)gray;
    a.H = H; a.W = W; a.n_images = n_images;
    a.spitch = spitch; a.dpitch = dpitch; a.sstride = src_image_stride; a.dstride = dst_image_stride;
    a.ry = pattern >= 2; a.rx = pattern == 1 || pattern == 2;   // red sites: BG (0,0), GB (0,1), RG (1,1), GR (1,0)
    a.cb = gray_shift == 14 ? 1868u : 3735u; a.cg = gray_shift == 14 ? 9617u : 19235u; a.cr = gray_shift == 14 ? 4899u : 9798u;
    a.shift = gray_shift;
    return 0;
}

int mocap_bayer_gray_u8(mocap_ctx_t c, const void* bayer, void* gray, int n_images, int H, int W, long spitch, long dpitch,
                        size_t src_image_stride, size_t dst_image_stride, int pattern, int gray_shift, void* stream)
{
    if (!c) return fail(MOCAP_E_INVALID, "null argument");
    BayerArgs a;
    int rc = bayer_args(a, bayer, gray, n_images, H, W, spitch, dpitch, src_image_stride, dst_image_stride, pattern, gray_shift);
    if (rc) return rc;
    if (set_device(c)) return MOCAP_E_HIP;
    launch_bayer_gray(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

int mocap_demosaic_u8(mocap_ctx_t c, const void* bayer, void* bgr, int H, int W, int spitch, void* stream)
{
    if (!c || !bayer || !bgr) return fail(MOCAP_E_INVALID, "null argument");
    if (H < 1 || W < 1 || spitch < W) return fail(MOCAP_E_INVALID, "bad geometry");
    if (set_device(c)) return MOCAP_E_HIP;
    launch_demosaic((const uint8_t*)bayer, (uint8_t*)bgr, H, W, spitch, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

// ---- the exchange: one all-gather of centroid records (RCCL over xGMI) -------------------------------------------
int mocap_comm_unique_id(void* id_out)
{
    if (!id_out) return fail(MOCAP_E_INVALID, "null argument");
    int rc = load_rccl();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id_out));
    return MOCAP_OK;
}

int mocap_comm_available(void)
{
    return load_rccl();
}

int mocap_comm_init(mocap_ctx_t c, const void* id, int rank, int world)
{
    if (!c || !id) return fail(MOCAP_E_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(MOCAP_E_INVALID, "rank %d of %d", rank, world);
    if (c->comm) return fail(MOCAP_E_STATE, "the context already has a communicator");
    int rc = load_rccl();
    if (rc) return rc;
    if (set_device(c)) return MOCAP_E_HIP;
    auto sc = std::make_shared<SharedComm>();
    HIP_TRY(hipEventCreateWithFlags(&sc->last, hipEventDisableTiming));
    IdBytes idb;
    memcpy(idb.internal, id, sizeof(idb.internal));
    int r_ = g_rccl.CommInitRank(&sc->comm, world, idb, rank);
    if (r_ != 0) {
        (void)hipEventDestroy(sc->last);
        return fail(MOCAP_E_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r_));
    }
    sc->rank = rank; sc->world = world; sc->device = c->device;
    sc->destroy_comm = [](void* comm) { return g_rccl.lib && g_rccl.CommDestroy(comm) == 0; };
    c->comm = sc;
    return MOCAP_OK;
}

int mocap_comm_share(mocap_ctx_t dst, mocap_ctx_t src)
{
    if (!dst || !src) return fail(MOCAP_E_INVALID, "null context");
    if (!src->comm) return fail(MOCAP_E_STATE, "the source context has no communicator");
    if (dst->comm) return fail(MOCAP_E_STATE, "the context already has a communicator");
    if (dst->device != src->device) return fail(MOCAP_E_INVALID, "contexts on different devices cannot share a communicator");
    dst->comm = src->comm;
    return MOCAP_OK;
}

int mocap_comm_destroy(mocap_ctx_t c)
{
    if (!c) return fail(MOCAP_E_INVALID, "null context");
    if (!c->comm) return MOCAP_OK;
    std::shared_ptr<SharedComm> sc = c->comm;
    c->comm.reset();
    if (sc.use_count() > 1) return MOCAP_OK; // other contexts of this rank still use it
    // the last user: destroyed here so that a failure can be reported (the destructor would do the same silently)
    (void)hipSetDevice(sc->device);
    if (sc->have_last) (void)hipEventSynchronize(sc->last);
    (void)hipEventDestroy(sc->last);
    sc->last = nullptr;
    void* comm = sc->comm;
    sc->comm = nullptr;
    if (comm && g_rccl.lib) RCCL_TRY(g_rccl.CommDestroy(comm));
    return MOCAP_OK;
}

int mocap_allgather_centroids(mocap_ctx_t c, const int32_t* local_records, int32_t* gathered, long ints_per_rank, void* stream)
{
    if (!c || !local_records || !gathered) return fail(MOCAP_E_INVALID, "null argument");
    if (ints_per_rank < 1) return fail(MOCAP_E_INVALID, "ints_per_rank = %ld", ints_per_rank);
    if (!c->comm) return fail(MOCAP_E_STATE, "mocap_comm_init / mocap_comm_share was not called for this context");
    if (set_device(c)) return MOCAP_E_HIP;
    SharedComm& sc = *c->comm;
    std::lock_guard<std::mutex> lk(sc.mu);
    if (sc.have_last) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, sc.last, 0)); // after the communicator's previous all-gather
    RCCL_TRY(g_rccl.AllGather(local_records, gathered, (size_t)ints_per_rank, 2 /* ncclInt32 */, sc.comm, (hipStream_t)stream));
    HIP_TRY(hipEventRecord(sc.last, (hipStream_t)stream));
    sc.have_last = true;
    return MOCAP_OK;
}

// ---- geometry stage --------------------------------------------------------------------------------------------
int mocap_correspond(mocap_ctx_t c, const void* pts, long pt_st, long pt_sc, const int32_t* counts, long cnt_st, long cnt_sc,
                     int pts_f64, int T, int C, int P, double cutoff,
                     int max_groups, double* root_xyz, double* root_err, double* root_grp, int32_t* root_idx,
                     int32_t* order, int32_t* n_roots, void* stream)
{
    if (!c || !pts || !counts || !root_xyz || !root_err || !root_grp || !root_idx || !order || !n_roots)
        return fail(MOCAP_E_INVALID, "null argument");
    if ((pt_st | pt_sc) & 1) return fail(MOCAP_E_INVALID, "point strides must be even (whole points)");
    if (T < 1 || C < 1 || C > 32 || P < 1 || P > 255 || max_groups < 1) return fail(MOCAP_E_INVALID, "T=%d C=%d P=%d max_groups=%d", T, C, P, max_groups);
    if (c->n_cam < C) return fail(MOCAP_E_STATE, "mocap_set_cameras: %d cameras set, %d needed", c->n_cam, C);
    if (c->n_F < C - 1) return fail(MOCAP_E_STATE, "mocap_set_fundamentals: %d matrices set, %d needed", c->n_F, C - 1);
    if (set_device(c)) return MOCAP_E_HIP;
    if (!correspond_fits(P, C)) // the plan is adaptive (geom.hip: corr_lds_plan); what is left are the candidate lists, P * C * 16 bytes
        return fail(MOCAP_E_UNSUPPORTED, "P=%d points x C=%d cameras needs %zu bytes of LDS per time step", P, C, correspond_smem_bytes(P, C));
    // error scratch: the groups of one time step lie back to back, so a step needs room for its total, not P x max_groups;
    // a step with more than max(2 * max_groups, 8192) groups in all reports MOCAP_CORR_E_GROUPS
    size_t budget = 2 * (size_t)max_groups > 8192 ? 2 * (size_t)max_groups : 8192;
    if (c->tune.corr_step_groups > 0) budget = (size_t)c->tune.corr_step_groups; // mocap_set_tuning(ctx, "corr_step_groups", n)
    if (budget > (size_t)P * max_groups) budget = (size_t)P * max_groups;
    if (budget > 0x7fffffff) budget = 0x7fffffff;
    size_t need = (size_t)T * budget;
    if (need > c->scratch_elems) {
        std::lock_guard<std::mutex> lk(c->mu);
        if (c->scratch) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->scratch)); c->scratch = nullptr; c->scratch_elems = 0; }
        HIP_TRY(hipMalloc(&c->scratch, sizeof(double) * need));
        c->scratch_elems = need;
    }
    CorrArgs a;
    a.cams = c->cams; a.pts = pts; a.counts = counts; a.pts_f64 = pts_f64; a.T = T; a.C = C; a.P = P;
    a.pt_st = pt_st; a.pt_sc = pt_sc; a.cnt_st = cnt_st; a.cnt_sc = cnt_sc;
    a.cutoff = cutoff; a.max_groups = max_groups; a.root_xyz = root_xyz; a.root_err = root_err; a.root_grp = root_grp;
    a.root_idx = root_idx; a.order = order; a.n_roots = n_roots; a.scratch = c->scratch; a.step_budget = (int)budget;
    a.prio = c->tune.corr_prio; // A/B switch (no effect measured)
    a.threads = c->tune.corr_threads;
    EvPair p; bool on;
    prof_begin(c, 2, (hipStream_t)stream, p, on);
    launch_correspond(a, (hipStream_t)stream);
    prof_end(c, 2, (hipStream_t)stream, p, on);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

int mocap_epipolar_scores(mocap_ctx_t c, const void* roots, int n_roots, const void* cand, int n_cand, int pts_f64, int f_index,
                          double* dist, float* lines, void* stream)
{
    if (!c || !roots || !cand || !dist) return fail(MOCAP_E_INVALID, "null argument");
    if (n_roots < 1 || n_cand < 1 || (long long)n_roots * n_cand > 0x7fffffffLL) return fail(MOCAP_E_INVALID, "n_roots=%d n_cand=%d", n_roots, n_cand);
    if (f_index < 0 || f_index >= c->n_F) return fail(MOCAP_E_STATE, "mocap_set_fundamentals: %d matrices set, index %d asked", c->n_F, f_index);
    if (set_device(c)) return MOCAP_E_HIP;
    EpiArgs a{c->cams, roots, cand, n_roots, n_cand, pts_f64, f_index, dist, lines};
    launch_epipolar_scores(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

int mocap_ba_residuals(mocap_ctx_t c, const double* params_host, int B, const double* pts, const uint8_t* valid, int N, int C,
                       float* residuals_host, int32_t* counts_host, void* stream)
{
    if (!c || !params_host || !pts || !valid || !residuals_host || !counts_host) return fail(MOCAP_E_INVALID, "null argument");
    if (B < 1 || N < 1 || C < 2 || C > 32 || (long long)B * N > (1LL << 28)) return fail(MOCAP_E_INVALID, "B=%d N=%d C=%d", B, N, C);
    if (c->n_cam < C) return fail(MOCAP_E_STATE, "mocap_set_cameras: %d cameras set, %d needed (their K and dist are used)", c->n_cam, C);
    if (set_device(c)) return MOCAP_E_HIP;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t np_ = (size_t)B * 6 * (C - 1), pbytes = (sizeof(double) * np_ + 15) & ~(size_t)15;
    const size_t rbytes = (sizeof(float) * (size_t)B * N + 15) & ~(size_t)15, need = pbytes + rbytes + sizeof(int32_t) * B;
    if (need > c->ba_pinned_bytes) {
        if (c->ba_pinned) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipHostFree(c->ba_pinned)); c->ba_pinned = nullptr; c->ba_pinned_bytes = 0; }
        HIP_TRY(hipHostMalloc(&c->ba_pinned, need * 2));
        c->ba_pinned_bytes = need * 2;
    }
    if ((size_t)B * N * 3 > c->ba_obj_elems) {
        if (c->ba_obj) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->ba_obj)); c->ba_obj = nullptr; c->ba_obj_elems = 0; }
        HIP_TRY(hipMalloc(&c->ba_obj, sizeof(double) * (size_t)B * N * 3 * 2));
        c->ba_obj_elems = (size_t)B * N * 3 * 2;
    }
    char* const pin = (char*)c->ba_pinned;
    memcpy(pin, params_host, sizeof(double) * np_);
    BaArgs a{c->cams, (const double*)pin, pts, valid, N, C, B, c->ba_obj, (float*)(pin + pbytes), (int32_t*)(pin + pbytes + rbytes)};
    launch_ba_residuals(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); // the one wait of an evaluation: the kernel has written the pinned block
    memcpy(residuals_host, a.res, sizeof(float) * (size_t)B * N);
    memcpy(counts_host, a.counts, sizeof(int32_t) * B);
    return MOCAP_OK;
}

int mocap_triangulate_batch(mocap_ctx_t c, const double* pts, const uint8_t* valid, int N, int C, int compact_k, double* xyz,
                            int32_t* ok, void* stream)
{
    if (!c || !pts || !valid || !xyz || !ok) return fail(MOCAP_E_INVALID, "null argument");
    if (N < 1 || C < 1 || C > 32) return fail(MOCAP_E_INVALID, "N=%d C=%d", N, C);
    if (c->n_cam < C) return fail(MOCAP_E_STATE, "mocap_set_cameras: %d cameras set, %d needed", c->n_cam, C);
    if (set_device(c)) return MOCAP_E_HIP;
    TriArgs a{c->cams, pts, valid, N, C, compact_k, xyz, ok};
    launch_triangulate(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

int mocap_reproject_batch(mocap_ctx_t c, const double* pts, const uint8_t* valid, const double* xyz, int N, int C, int compact_k,
                          double* mse, int32_t* ok, void* stream)
{
    if (!c || !pts || !valid || !xyz || !mse || !ok) return fail(MOCAP_E_INVALID, "null argument");
    if (N < 1 || C < 1 || C > 32) return fail(MOCAP_E_INVALID, "N=%d C=%d", N, C);
    if (c->n_cam < C) return fail(MOCAP_E_STATE, "mocap_set_cameras: %d cameras set, %d needed", c->n_cam, C);
    if (set_device(c)) return MOCAP_E_HIP;
    ReprojArgs a{c->cams, pts, valid, xyz, N, C, compact_k, mse, ok};
    launch_reproject(a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return MOCAP_OK;
}

} // extern "C"
