// rto_api.hip -- implementation of the C ABI in include/rto_hip.h (librto_hip.so).
// Host side: argument checks, the canonical-octree repack, uniform hoisting
// (inverse(view), tan(fov/2), frustum planes) and kernel launches.  No torch, no
// oracle, no CPU fallback: every entry point needs a gfx950 device.
#include <chrono>
#include "rto_device.hip.h"
static int lean_block(int path);

#include "../host/rtmath.h"

#include <algorithm>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace rto;

struct rto_context {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    std::string deviceName;

    // resident octree
    rto_node* d_nodes = nullptr;
    int64_t numNodes = 0;
    uint2* d_desc = nullptr;
    int* d_descFirstChild = nullptr;
    int64_t numInternal = 0;
    int rootSize = 0, depth = 0;
    bool canonical = false;
    bool exactGridAllowed = true;   // rto_debug_set_exact_grid(0): the general 12-plane child test whatever the grid (tests, A/B)
    bool exactGrid = false;         // every node plane gridMin + k * voxelSize (k = 0 .. rootSize) is computed without rounding: see grid_is_exact
    float gridMin[3] = { 0, 0, 0 };
    float voxelSize = 1.f;
    int kernelMode = RTO_KERNEL_AUTO;
    float solidCentre[3] = { 0, 0, 0 };   // centre of the bounding box of the solid leaves, voxel units (launch-order heuristic)
    int solidLo[3] = { 0, 0, 0 }, solidHi[3] = { 0, 0, 0 };   // that bounding box, voxel units; lo > hi: no solid leaf at all

    // frustum culling state
    bool culling = false;
    int rootVisible = 1;
    uint8_t* d_vis = nullptr;
    int* d_remap = nullptr;
    int* d_blockCount = nullptr;
    int* d_blockBase = nullptr;
    int64_t* d_visibleCount = nullptr;
    rto_node* d_compact = nullptr;
    bool compactValid = false;       // d_compact / d_remap hold the compaction of the CURRENT visibility flags (made on demand)
    bool otherStreams = false;       // a frame was launched on a stream other than c->stream since the last device-wide wait
    bool foreignCaptured = false;    // a frame was CAPTURED on another stream: its replays are the caller's, every later update waits for the device
    int64_t visibleNodes = 0;
    // canonical trees: the update is one kernel and reads nothing back (k_cull_desc); where traversals start and how many nodes
    // survived stay on the device (d_start) and reach the host only when somebody asks (sync_cull_state)
    int4* d_descPos = nullptr;       // position + size of every internal node, descriptor order
    int* d_cullBlockCount = nullptr; // k_cull_desc: per-block partials
    int* d_cullBlockFirst = nullptr;
    StartState* d_start = nullptr;
    bool cullAsync = false;          // the flags in force come from k_cull_desc
    bool visAllOnes = false;         // d_vis, the descriptors' visibility bits and d_start say "every node visible" (rto_update_frustum's host-side proof left them so)
    bool cullShortcut = true;        // rto_debug_set_frustum_shortcut
    bool lastUpdateProven = false;   // the last rto_update_frustum was answered by the host-side proof (rto_debug_last_frustum_update_proven)
    bool cullStateStale = false;     // rootVisible / visibleNodes below are older than d_start
    bool cullCaptured = false;       // an update was stream-captured: replays change d_start behind the host's back
    hipEvent_t evCull = nullptr;     // recorded on c->stream behind the last kernel that rewrote the visibility state (descriptor bits, d_vis,
    unsigned long cullSeq = 0;       // d_start) when a frame on ANOTHER stream first needs it (order_after_cull); cullSeq counts the rewrites,
    unsigned long evCullSeq = 0;     // evCullSeq is the rewrite the event stands behind, evCullDone: the host has seen it complete
    bool evCullDone = false;

    // temporal launch order (packed kernel): an earlier frame's per-tile cost -> this frame's slot->tile table.
    // The tables are written and read by kernels in stream order, so every launch stream owns a set of its own:
    // frames in flight on different streams of one context never share (or race on) a table.
    int numCUs = 256;
    int residentW = 0, residentH = 0;   // size of the frame rto_render_resident left in d_frame
    int orderPolicy = 1;            // 0 = centre-out only, 1 = temporal (falls back to centre-out without history)
    int orderPeriod = 8;            // rebuild the table every orderPeriod-th frame: cost maps change slowly (tools/order_period.py:
                                    // 8 or 16 beat 4 for a static, an orbiting and a fast-moving camera alike)
    struct OrderState {
        int* d_tileCost = nullptr;      // tile -> trip count of the last colour / shade frame that rendered it (row-major over all tiles)
        // launch slot -> tile: a permutation of the tiles of `box`.  TWO tables: [0] for plain launches, [1] for launches that are
        // being stream-captured.  A graph replay re-runs its k_order_build nodes and rewrites its table behind the host's back, so
        // plain launches never read table [1]; and every capture starts with a rebuild node of its own (capId), so a replay never
        // depends on what plain launches, another graph or its own tail left in the table.
        struct Table { int* d = nullptr; bool valid = false; int box[4] = { 0, 0, 0, 0 }; int age = 0; int stretch = 1; };   // stretch: see prepare_schedule
        Table tab[2];
        int active = 0;                 // the table the launch being prepared uses
        unsigned long long capId = 0;   // id of the capture table [1] was last (re)built in
        int tiles = 0;                  // tile count the buffers are sized for
        long key[7] = { 0, 0, 0, 0, 0, 0, 0 };   // W, H, numParts, part, bandRows, tiles, path (0 octree / 1 triangles) of the frames the history belongs to
        bool costValid = false;         // a frame of this geometry has recorded its costs
        float lastCam[19] = { 0 };      // inverse view + eye of the last single-frame launch: a camera in motion rebuilds the table for every frame
        bool haveCam = false;
        bool fixed = false;             // debug: the caller supplied the table (over ALL tiles), do not rebuild it
        int* d_queue = nullptr;         // persistent-threads variant: the slot counter (zeroed in front of every launch)
        unsigned* d_tileMask = nullptr; // occupancy masks of the frames of one launch: kMaxBatch regions of maskWords words (mask_block)
        size_t maskWords = 0;           // strips * tilesX + 1 of the frame size the buffer was made for
        unsigned long lastUse = 0;      // orderClock value of the last launch on this stream (eviction order)
    };
    typedef std::pair<hipStream_t, int> OrderKey;       // launch stream, path (0 octree / 1 triangle / 2 nearest-hit frames): an application that
                                                         // renders two kinds of frame on one stream keeps a history for each
    std::map<OrderKey, OrderState> orders;
    OrderKey lastOrderKey = OrderKey(nullptr, -1);       // what the rto_debug_* order functions refer to
    static constexpr size_t kMaxOrderStreams = 16;   // a 17th stream evicts the least recently used entry
    unsigned long orderClock = 0;
    int* d_sortViolations = nullptr;            // k_sort_scatter: out-of-range writes refused (must stay 0; rto_debug_sort_violations)

    // occupancy mask: the coarse cells of the tree at level cellLevel, projected per frame by the launch's first workgroups (mask_block)
    int4* d_cells = nullptr;
    int numCells = 0, cellLevel = 0;
    unsigned maskStamp = 0;                     // one fresh value per frame launched
    int maskMode = 1;                           // rto_debug_set_tile_mask: 0 off, 1 on, 2 on + built by a launch of its own in front of the frame (tests)

    // voxels retained by rto_build_octree (so that rto_build_leaf_triangles can run without a second upload)
    uint8_t* d_vox = nullptr;
    int voxDim[3] = { 0, 0, 0 };
    unsigned long long* d_mcCases = nullptr;    // 256 packed Marching-Cubes cases

    // leaf triangles (config 5 extension)
    float* d_tris = nullptr;
    int* d_triOffset = nullptr;
    uint2* d_triRec = nullptr;       // k_trace_lean_triangles: one record per interesting child (k_unified_fill)
    int64_t numTris = 0;

    // separable ray terms (per column / per row), cached by (W, H, aspect, tanHalfFov)
    float* d_rayX = nullptr;
    float* d_rayY = nullptr;
    int rayW = 0, rayH = 0;
    float rayAspect = 0.f, rayTan = 0.f;

    // outputs / instrumentation
    float4* d_frame = nullptr;
    size_t frameCap = 0;
    int* d_steps = nullptr;
    size_t stepsCap = 0;
    Counters* d_counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool eventsOff = false;    // rto_timing_begin(ctx, -1): no hipEventRecord around the traversal kernels at all
    // optional ring of event pairs, one per traversal-kernel launch (rto_timing_begin / rto_timing_read)
    std::vector<hipEvent_t> ringStart, ringStop;
    size_t ringUsed = 0;
    hipEvent_t lastA = nullptr, lastB = nullptr;
    float buildMs = -1.f;      // device time of the last rto_build_octree (pyramid + emission kernels)
    bool descPooled = false;   // d_descFirstChild lives in d_desc's allocation (Morton-order build)
    bool asyncPooled = false;  // d_nodes / d_desc come from the stream-ordered memory pool (hipMallocAsync): a rebuild reuses them without hipMalloc
    int buildPath = 0;         // 0 = automatic (Morton-order build where it applies), 1 = level-by-level build (rto_debug_set_build_path)
    float buildUploadMs = -1.f;
};

static thread_local std::string g_createError;

static int fail(rto_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_createError = msg;
    return code;
}

#define RTO_HIP(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctx, RTO_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

// A/B knobs and developer aids read from the environment exist only in builds made with -DRTO_DEV_KNOBS (tools/build_variants.sh):
// the shipped library reads NO environment variable -- a stray RTO_* in a user's environment cannot change what it does.
static inline long long dev_env(const char* name, long long unset) {
#if defined(RTO_DEV_KNOBS)
    const char* e = std::getenv(name);
    return (e && *e) ? std::atoll(e) : unset;
#else
    (void)name;
    return unset;
#endif
}

// Test hook (rto_debug_fault_alloc(k), an explicit call: never the environment): the k-th (1-based) buffer allocation of the
// frustum-update buffers / rto_comm buffers from now on fails, so that the all-or-nothing clean-up below can be exercised
// (tests/test_gpu_parity.py, tests/_fault_alloc_worker.py).  0: never.
static long g_faultCountdown = 0;
static hipError_t fallible_malloc(void** p, size_t bytes) {
    if (g_faultCountdown > 0 && --g_faultCountdown == 0) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
extern "C" int rto_debug_fault_alloc(long k) { g_faultCountdown = k > 0 ? k : 0; return RTO_OK; }

static void free_cull_buffers(rto_context* c) {
    (void)hipFree(c->d_cullBlockCount); c->d_cullBlockCount = nullptr;
    (void)hipFree(c->d_cullBlockFirst); c->d_cullBlockFirst = nullptr;
    c->cullAsync = false; c->cullStateStale = false; c->cullCaptured = false; c->visAllOnes = false;
    (void)hipFree(c->d_vis); c->d_vis = nullptr;
    (void)hipFree(c->d_remap); c->d_remap = nullptr;
    (void)hipFree(c->d_blockCount); c->d_blockCount = nullptr;
    (void)hipFree(c->d_blockBase); c->d_blockBase = nullptr;
    (void)hipFree(c->d_compact); c->d_compact = nullptr;
}

static void free_octree(rto_context* c) {
    if (c->asyncPooled) {
        (void)hipDeviceSynchronize();          // like hipFree: frames on caller streams may still read the arrays
        if (c->d_nodes) (void)hipFreeAsync(c->d_nodes, c->stream);
        if (c->d_desc) (void)hipFreeAsync(c->d_desc, c->stream);
    }
    else { (void)hipFree(c->d_nodes); (void)hipFree(c->d_desc); }
    c->d_nodes = nullptr; c->d_desc = nullptr; c->asyncPooled = false;
    if (!c->descPooled) (void)hipFree(c->d_descFirstChild);
    c->d_descFirstChild = nullptr; c->descPooled = false;
    free_cull_buffers(c);
    (void)hipFree(c->d_descPos); c->d_descPos = nullptr;
    (void)hipFree(c->d_cells); c->d_cells = nullptr; c->numCells = 0; c->cellLevel = 0;
    (void)hipFree(c->d_tris); c->d_tris = nullptr;
    (void)hipFree(c->d_triOffset); c->d_triOffset = nullptr;
    (void)hipFree(c->d_triRec); c->d_triRec = nullptr;
    (void)hipFree(c->d_vox); c->d_vox = nullptr;
    c->voxDim[0] = c->voxDim[1] = c->voxDim[2] = 0;
    c->numTris = 0;
    for (auto& kv : c->orders) { kv.second.tab[0].valid = kv.second.tab[1].valid = false; kv.second.costValid = false; }
    c->numNodes = c->numInternal = 0;
    c->canonical = false; c->culling = false; c->rootVisible = 1; c->visibleNodes = 0;
}

extern "C" {

int rto_create(int device_ordinal, rto_context** out) {
    if (!out) return fail(nullptr, RTO_E_INVALID, "rto_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, RTO_E_NO_DEVICE, std::string("rto_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(nullptr, RTO_E_NO_DEVICE, "rto_create: device ordinal out of range");
    rto_context* c = new rto_context();
    c->device = device_ordinal;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_ordinal)) != hipSuccess) {
        delete c;
        return fail(nullptr, RTO_E_HIP, std::string("rto_create: ") + hipGetErrorString(e));
    }
    c->deviceName = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {   // keep freed build scratch in the stream-ordered pool (see BuildScratch)
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device_ordinal) == hipSuccess && pool) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
    }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        std::string msg = "rto_create: device is " + c->deviceName + ", this library carries gfx950 code only";
        delete c;
        return fail(nullptr, RTO_E_NO_DEVICE, msg);
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->evCull, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc(&c->d_counters, sizeof(Counters))) != hipSuccess ||
        (e = hipMalloc(&c->d_visibleCount, 2 * sizeof(int64_t))) != hipSuccess ||     // [0] count of visible nodes, [1] the root's flag
        (e = hipMalloc(&c->d_start, sizeof(StartState))) != hipSuccess || (e = hipMemset(c->d_start, 0, sizeof(StartState))) != hipSuccess ||
        (e = hipMalloc(&c->d_sortViolations, sizeof(int))) != hipSuccess || (e = hipMemset(c->d_sortViolations, 0, sizeof(int))) != hipSuccess) {
        std::string msg = std::string("rto_create: ") + hipGetErrorString(e);
        rto_destroy(c);
        return fail(nullptr, RTO_E_HIP, msg);
    }
    *out = c;
    return RTO_OK;
}

void rto_destroy(rto_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_octree(c);
    (void)hipFree(c->d_frame);
    (void)hipFree(c->d_rayX);
    (void)hipFree(c->d_rayY);
    for (auto& kv : c->orders) { (void)hipFree(kv.second.d_tileCost); (void)hipFree(kv.second.tab[0].d); (void)hipFree(kv.second.tab[1].d); (void)hipFree(kv.second.d_queue); (void)hipFree(kv.second.d_tileMask); }
    c->orders.clear();
    (void)hipFree(c->d_mcCases);
    (void)hipFree(c->d_steps);
    (void)hipFree(c->d_counters);
    (void)hipFree(c->d_visibleCount);
    (void)hipFree(c->d_start);
    (void)hipFree(c->d_sortViolations);
    for (hipEvent_t e : c->ringStart) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ringStop) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evCull) (void)hipEventDestroy(c->evCull);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* rto_last_error(const rto_context* ctx) { return ctx ? ctx->err.c_str() : g_createError.c_str(); }

int rto_device_name(const rto_context* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf || buflen == 0) return RTO_E_INVALID;
    std::snprintf(buf, buflen, "%s", ctx->deviceName.c_str());
    return RTO_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- canonical repack
// A flat array is canonical when it is what setOctree's BFS emits for a tree built by
// createOctreeFromVoxelGrid: root (0,0,0,2^d) at index 0; every internal node has its 8
// children at 8 consecutive indices, in child order, each with the derived position/size.
static inline bool is_terminal(const rto_node& n) { return n.isUniform == 1 || n.isLeaf == 1; }

static bool build_descriptors(const rto_node* nodes, int64_t n, std::vector<uint2>& desc,
                              std::vector<int>& firstChild, int& rootSize, int& depth) {
    desc.clear(); firstChild.clear();
    if (n <= 0) return false;
    const rto_node& root = nodes[0];
    rootSize = root.size; depth = 0;
    if (is_terminal(root) || root.x != 0 || root.y != 0 || root.z != 0) return false;
    if (root.size < 2 || (root.size & (root.size - 1)) != 0) return false;
    while ((1 << depth) < root.size) depth++;
    if (depth > kMaxDepth) return false;
    std::vector<int> rank((size_t)n + 1);   // rank[i] = internal nodes among [0, i)
    int r = 0;
    for (int64_t i = 0; i < n; i++) { rank[(size_t)i] = r; r += is_terminal(nodes[i]) ? 0 : 1; }
    rank[(size_t)n] = r;
    desc.resize((size_t)r); firstChild.resize((size_t)r);
    for (int64_t i = 0; i < n; i++) {
        const rto_node& nd = nodes[i];
        if (is_terminal(nd)) continue;
        if (nd.size < 2 || (nd.size & (nd.size - 1)) != 0) return false;
        const int half = nd.size / 2;
        const int64_t c0 = nd.child[0];
        if (c0 <= i || c0 + 7 >= n) return false;
        unsigned imask = 0, smask = 0;
        for (int k = 0; k < 8; k++) {
            if (nd.child[k] != c0 + k) return false;
            const rto_node& ch = nodes[c0 + k];
            if (ch.size != half || ch.x != nd.x + ((k & 1) ? half : 0) || ch.y != nd.y + ((k & 2) ? half : 0) ||
                ch.z != nd.z + ((k & 4) ? half : 0))
                return false;
            if (!is_terminal(ch)) imask |= 1u << k;
            else if (ch.isSolid == 1) smask |= 1u << k;
        }
        const int d = rank[(size_t)i];
        desc[(size_t)d] = make_uint2(smask | (imask << 8) | 0xff0000u, (unsigned)rank[(size_t)c0]);
        firstChild[(size_t)d] = (int)c0;
    }
    return true;
}

static void bounds_from_nodes(const rto_node* nodes, int64_t n, const float grid_min[3], float voxel_size, rto_scene_bounds* b);   // rto_split.inc

static int build_cells(rto_context* c);

extern "C" {

int rto_upload_octree(rto_context* c, const rto_node* nodes, int64_t n, const float grid_min[3], float voxel_size) {
    if (!c) return RTO_E_INVALID;
    if (!nodes || n <= 0 || !grid_min) return fail(c, RTO_E_INVALID, "rto_upload_octree: empty node array");
    if (n > 0x7fffffff) return fail(c, RTO_E_INVALID, "rto_upload_octree: node indices are int32 (GPUNodes.child)");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    free_octree(c);
    std::memcpy(c->gridMin, grid_min, sizeof c->gridMin);
    c->voxelSize = voxel_size;
    c->numNodes = n;
    c->visibleNodes = n;
    RTO_HIP(c, hipMalloc(&c->d_nodes, (size_t)n * sizeof(rto_node)));
    RTO_HIP(c, hipMemcpy(c->d_nodes, nodes, (size_t)n * sizeof(rto_node), hipMemcpyHostToDevice));

    {   // where the geometry is: tiles nearest its projection are launched first (heavy waves early)
        rto_scene_bounds sb;
        bounds_from_nodes(nodes, n, grid_min, voxel_size, &sb);
        const bool any = sb.solid_lo[0] <= sb.solid_hi[0];
        for (int a = 0; a < 3; a++) {
            c->solidCentre[a] = any ? 0.5f * (float)((long)sb.solid_lo[a] + (long)sb.solid_hi[a]) : 0.5f * (float)nodes[0].size;
            c->solidLo[a] = sb.solid_lo[a]; c->solidHi[a] = sb.solid_hi[a];
        }
    }
    std::vector<uint2> desc;
    std::vector<int> firstChild;
    c->canonical = build_descriptors(nodes, n, desc, firstChild, c->rootSize, c->depth);
    if (c->canonical) {
        c->numInternal = (int64_t)desc.size();
        RTO_HIP(c, hipMalloc(&c->d_desc, desc.size() * sizeof(uint2)));
        RTO_HIP(c, hipMemcpy(c->d_desc, desc.data(), desc.size() * sizeof(uint2), hipMemcpyHostToDevice));
        RTO_HIP(c, hipMalloc(&c->d_descFirstChild, firstChild.size() * sizeof(int)));
        RTO_HIP(c, hipMemcpy(c->d_descFirstChild, firstChild.data(), firstChild.size() * sizeof(int), hipMemcpyHostToDevice));
    } else {
        int64_t internal = 0;
        for (int64_t i = 0; i < n; i++) internal += is_terminal(nodes[i]) ? 0 : 1;
        c->numInternal = internal;
        c->rootSize = nodes[0].size;
    }
    return build_cells(c);
}

}  // extern "C"

// ---------------------------------------------------------------- N4: build on the GPU
namespace {
struct LevelBuf {
    int4* coords = nullptr;
    uint8_t* state = nullptr;
    uint8_t* flag = nullptr;
    int* rank = nullptr;
    int64_t m = 0;       // nodes at this level
    int64_t k = 0;       // internal nodes at this level
};
// Scratch of the build entry points: stream-ordered allocations from the device's default memory pool (its release
// threshold is raised in rto_create, so a second build reuses the memory of the first instead of calling hipMalloc
// ~50 times); everything is returned to the pool, in stream order, when the build leaves scope.
struct BuildScratch {
    hipStream_t stream;
    std::vector<void*> allocs;
    explicit BuildScratch(hipStream_t s) : stream(s) {}
    ~BuildScratch() { for (void* p : allocs) (void)hipFreeAsync(p, stream); }
    template <class T> hipError_t alloc(T** p, size_t count) {
        hipError_t e = hipMallocAsync(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T), stream);
        if (e == hipSuccess) allocs.push_back(*p);
        return e;
    }
};
}  // namespace

// Per-octree derived data of canonical trees, made once after upload / build on the context's stream (synchronises):
// descPos (position + size of every internal node: the frustum update and the cells read it) and the coarse cells of the
// occupancy mask -- the deepest level whose cells (internal nodes at that depth + solid leaves at or above it) number at most
// kMaskMaxCells.
constexpr int kMaskMaxCells = 8192;      // config 2: 5,624 cells (depth 5) serve as well as 20,504 (depth 6): 38.7 us either way; projecting them costs a quarter
// Does the reference's node-box arithmetic (S/RT:265-266: nodeMin = gridMin + vec3(x, y, z) * voxelSize, nodeMax = nodeMin + vec3(size) *
// voxelSize, in float) round anywhere on this grid?  If for every k = 0 .. rootSize the product k * voxelSize and the sum gridMin + that
// product are exact, every plane of every node is the real number gridMin + k * voxelSize, and a child's max plane (min + size *
// voxelSize) IS its upper sibling's min plane, bit for bit: the lean kernels then compute 9 plane parameters per trip instead of 12
// (child_axis_terms_exact) -- the same floats by construction.  True for power-of-two voxel sizes on "round" origins (the test spheres:
// -0.5, 2^-8) and for integer grids (sceneCache.bin: origin (-2125, -1215, -150), voxel 10); any other grid takes the general form.
static bool grid_is_exact(const float gridMin[3], float voxelSize, int rootSize) {
    if (!(voxelSize > 0.0f) || !std::isfinite(voxelSize) || rootSize <= 0 || rootSize > (1 << 20)) return false;
    for (int a = 0; a < 3; a++) {
        if (!std::isfinite(gridMin[a])) return false;
        for (int k = 0; k <= rootSize; k++) {
            const volatile float prod = (float)k * voxelSize;               // the kernels' own operations, one rounding each
            const volatile float sum = gridMin[a] + prod;
            if ((double)prod != (double)k * (double)voxelSize) return false;
            if ((double)sum != (double)gridMin[a] + (double)k * (double)voxelSize) return false;
        }
    }
    return true;
}

static int build_cells_impl(rto_context* c);
static int build_cells(rto_context* c) {
    c->exactGrid = c->exactGridAllowed && grid_is_exact(c->gridMin, c->voxelSize, c->rootSize);
    static const bool trace = dev_env("RTO_BUILD_TRACE", 0) != 0;            // developer aid (dev builds): host time of this step, on stderr
    if (!trace) return build_cells_impl(c);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = build_cells_impl(c);
    std::fprintf(stderr, "[rto] derived data of the octree (descPos, occupancy cells): %.3f ms of host time\n",
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}
static int build_cells_impl(rto_context* c) {
    (void)hipFree(c->d_descPos); c->d_descPos = nullptr;
    (void)hipFree(c->d_cells); c->d_cells = nullptr; c->numCells = 0; c->cellLevel = 0;
    if (!c->canonical || c->numInternal <= 0) return RTO_OK;
    hipStream_t s = c->stream;
    const int nbInt = (int)((c->numInternal + kBlock - 1) / kBlock);
    RTO_HIP(c, hipMalloc(&c->d_descPos, (size_t)c->numInternal * sizeof(int4)));
    hipLaunchKernelGGL(k_desc_pos, dim3(nbInt), dim3(kBlock), 0, s, c->d_nodes, c->d_descFirstChild, c->numInternal, c->d_descPos);
    BuildScratch scratch(s);
    int* d_counts = nullptr;
    constexpr int kN = 2 * (kMaxDepth + 1);
    RTO_HIP(c, scratch.alloc(&d_counts, (size_t)kN + 1));                    // + the fill cursor
    RTO_HIP(c, hipMemsetAsync(d_counts, 0, (kN + 1) * sizeof(int), s));
    hipLaunchKernelGGL(k_cells_count, dim3(nbInt), dim3(kBlock), 0, s, c->d_desc, c->d_descPos, c->numInternal, c->depth, d_counts);
    RTO_HIP(c, hipGetLastError());
    int counts[kN];
    RTO_HIP(c, hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    int level = 0;
    long long solidAbove = 0, best = 0;
    for (int L = 1; L <= c->depth && L <= kMaxDepth; L++) {
        solidAbove += counts[kMaxDepth + 1 + L];
        const long long n = counts[L] + solidAbove;
        static const long long cellsEnv = dev_env("RTO_MASK_CELLS", 0);      // A/B knob (dev builds)
        if (n > (cellsEnv > 0 ? cellsEnv : (long long)kMaskMaxCells)) break;
        level = L; best = n;
    }
    if (level == 0 || best <= 0) return RTO_OK;                              // nothing solid, or already too many cells at depth 1
    RTO_HIP(c, hipMalloc(&c->d_cells, (size_t)best * sizeof(int4)));
    hipLaunchKernelGGL(k_cells_fill, dim3(nbInt), dim3(kBlock), 0, s, c->d_desc, c->d_descPos, c->numInternal, c->depth, level, c->d_cells, (int)best, d_counts + kN);
    RTO_HIP(c, hipGetLastError());
    int filled = 0;
    RTO_HIP(c, hipMemcpyAsync(&filled, d_counts + kN, sizeof filled, hipMemcpyDeviceToHost, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    if (filled != (int)best) return fail(c, RTO_E_HIP, "occupancy cells: count and fill disagree");
    c->numCells = (int)best; c->cellLevel = level;
    return RTO_OK;
}

// The four-launch build (k_mb_*): fills c->d_nodes / d_desc / d_descFirstChild from c->d_vox.  *total / *internal: sizes;
// solidBox: lo[3], hi[3] of the cells that hold FILLED voxels (level-1 cell precision; lo > hi: nothing solid).
// The voxels are uploaded here (between the events e0 and e1), after the scratch has been allocated and the chunk sums zeroed.
static int build_octree_morton(rto_context* c, hipStream_t s, BuildScratch& scratch, int R, int dimX, int dimY, int dimZ,
                               const uint8_t* voxels, hipEvent_t e0, hipEvent_t e1, int64_t* total, int64_t* internal, int solidBox[6]) {
    MbLevels Lv;
    std::memset(&Lv, 0, sizeof Lv);
    Lv.R = R; Lv.dimX = dimX; Lv.dimY = dimY; Lv.dimZ = dimZ;
    const int B = R < kMbBrickLevels ? R : kMbBrickLevels;
    const int bricksX = (dimX + 31) / 32, bricksY = (dimY + 31) / 32, bricksZ = (dimZ + 31) / 32;
    // bricks that do not touch the grid are never computed: their cells must read EMPTY with no mixed children
    const bool allBricks = R < kMbBrickLevels || ((long long)bricksX << 5 >= (1ll << R) && (long long)bricksY << 5 >= (1ll << R) && (long long)bricksZ << 5 >= (1ll << R));
    long long scanCells = 0, emitCells = 0;
    for (int l = 1; l <= R; l++) {
        const size_t cells = (size_t)1 << (3 * (R - l));
        RTO_HIP(c, scratch.alloc(&Lv.state[l], cells));
        if (!allBricks && l <= B) RTO_HIP(c, hipMemsetAsync(Lv.state[l], 0, cells, s));
        Lv.emitOffset[l] = emitCells; emitCells += l == R ? 1 : (long long)(cells / 8);      // k_mb_emit: one thread per group of 8 siblings
        if (l >= 2) {
            RTO_HIP(c, scratch.alloc(&Lv.cnt[l], cells));
            RTO_HIP(c, scratch.alloc(&Lv.group[l], cells));
            if (!allBricks && l <= B) RTO_HIP(c, hipMemsetAsync(Lv.cnt[l], 0, cells, s));
            Lv.cntOffset[l] = scanCells;
            scanCells += (long long)((cells + kMbChunk - 1) / kMbChunk) * kMbChunk;
        }
    }
    Lv.cntOffset[R + 1] = scanCells; Lv.emitOffset[R + 1] = emitCells;
    if (R >= 2) Lv.cntOffset[1] = 0;
    const unsigned chunks = (unsigned)(scanCells / kMbChunk);
    int* d_chunk = nullptr;
    int* d_brickBox = nullptr;
    MbTables* d_tab = nullptr;
    const int runsX = (bricksX + kMbRun - 1) / kMbRun;              // a block takes kMbRun bricks along x
    const int numBricks = runsX * bricksY * bricksZ;                // = blocks of k_mb_bricks
    RTO_HIP(c, scratch.alloc(&d_chunk, (size_t)chunks + 1));
    RTO_HIP(c, scratch.alloc(&d_brickBox, (size_t)numBricks * 6));
    RTO_HIP(c, scratch.alloc(&d_tab, 1));
    RTO_HIP(c, hipMemsetAsync(d_chunk, 0, ((size_t)chunks + 1) * sizeof(int), s));       // summed into by atomics
    RTO_HIP(c, hipEventRecord(e0, s));
    RTO_HIP(c, hipMemcpyAsync(c->d_vox, voxels, (size_t)dimX * dimY * dimZ, hipMemcpyHostToDevice, s));
    RTO_HIP(c, hipEventRecord(e1, s));
    hipLaunchKernelGGL(k_mb_bricks, dim3((unsigned)numBricks), dim3(kBlock), 0, s, c->d_vox, Lv, bricksX, runsX, bricksY, d_brickBox, d_chunk);
    hipLaunchKernelGGL(k_mb_top_scan, dim3(1), dim3(1024), 0, s, Lv, d_brickBox, numBricks, d_chunk, d_tab);
    RTO_HIP(c, hipGetLastError());
    MbTables tab;
    RTO_HIP(c, hipMemcpyAsync(&tab, d_tab, sizeof tab, hipMemcpyDeviceToHost, s));
    RTO_HIP(c, hipStreamSynchronize(s));          // the only read-back before the tree is complete: it sizes the outputs
    if (tab.total > 0x7fffffff) return fail(c, RTO_E_UNSUPPORTED, "rto_build_octree: node indices are int32 (GPUNodes.child)");
    *total = tab.total; *internal = tab.internal;
    {   // level-1 cells on the grid's boundary stick out of it: clip
        const int dims[3] = { dimX, dimY, dimZ };
        for (int a = 0; a < 3; a++) { solidBox[a] = tab.solidLo[a]; solidBox[3 + a] = std::min(tab.solidHi[a], dims[a]); }
    }
    // result arrays from the stream-ordered pool (its release threshold is raised in rto_create): the first build pays the
    // allocation, a rebuild of a similar scene gets the memory back without a hipMalloc (~0.1 ms each)
    c->asyncPooled = true;
    RTO_HIP(c, hipMallocAsync(reinterpret_cast<void**>(&c->d_nodes), (size_t)tab.total * sizeof(rto_node), s));
    if (tab.internal > 0) {
        // one allocation for both descriptor arrays: d_descFirstChild lives behind d_desc
        const size_t descBytes = ((size_t)tab.internal * sizeof(uint2) + 255) & ~(size_t)255;
        char* both = nullptr;
        RTO_HIP(c, hipMallocAsync(reinterpret_cast<void**>(&both), descBytes + (size_t)tab.internal * sizeof(int), s));
        c->d_desc = reinterpret_cast<uint2*>(both);
        c->d_descFirstChild = reinterpret_cast<int*>(both + descBytes);
        c->descPooled = true;
    }
    unsigned* d_cellOf = nullptr;                                  // descriptor index -> Morton index of the cell (its level follows from the bases)
    RTO_HIP(c, scratch.alloc(&d_cellOf, (size_t)tab.internal + 1));
    if (chunks) hipLaunchKernelGGL(k_mb_group_ranks, dim3(chunks), dim3(kBlock), 0, s, Lv, d_chunk, d_tab, d_cellOf);
    hipLaunchKernelGGL(k_mb_emit, dim3((unsigned)std::max<int64_t>(1, (tab.internal + kMbEmitCells - 1) / kMbEmitCells)), dim3(kBlock), 0, s,
                       c->d_vox, Lv, d_tab, d_cellOf, c->d_nodes, c->d_desc, c->d_descFirstChild);
    RTO_HIP(c, hipGetLastError());
    return RTO_OK;
}

extern "C" {

int rto_build_octree(rto_context* c, const uint8_t* voxels, int dimX, int dimY, int dimZ, const float grid_min[3], float voxel_size) {
    if (!c) return RTO_E_INVALID;
    if (!voxels || !grid_min || dimX <= 0 || dimY <= 0 || dimZ <= 0)
        return fail(c, RTO_E_INVALID, "rto_build_octree: empty voxel grid (createOctreeFromVoxelGrid returns no root for it)");
    int maxDim = dimX > dimY ? dimX : dimY;
    if (dimZ > maxDim) maxDim = dimZ;
    int R = 0;
    while ((1 << R) < maxDim) R++;
    if (R > kMaxDepth) return fail(c, RTO_E_UNSUPPORTED, "rto_build_octree: grid too large");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    free_octree(c);
    std::memcpy(c->gridMin, grid_min, sizeof c->gridMin);
    c->voxelSize = voxel_size;
    hipStream_t s = c->stream;
    BuildScratch scratch(c->stream);
    hipEvent_t e0, e1, e2;
    RTO_HIP(c, hipEventCreate(&e0)); RTO_HIP(c, hipEventCreate(&e1)); RTO_HIP(c, hipEventCreate(&e2));
    struct EvGuard { hipEvent_t a, b, d; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(d); } } evg{ e0, e1, e2 };

    // ---- voxels -> HBM
    const size_t nvox = (size_t)dimX * dimY * dimZ;
    uint8_t* d_vox = nullptr;
    RTO_HIP(c, hipMalloc(&d_vox, nvox));
    c->d_vox = d_vox; c->voxDim[0] = dimX; c->voxDim[1] = dimY; c->voxDim[2] = dimZ;    // kept: rto_build_leaf_triangles reads it
    int64_t total = 0, internal = 0;
    int box[6];
    bool haveBox = false;              // the Morton-order build delivers the solid box with its one read-back
    int* d_bbox = nullptr;
    RTO_HIP(c, scratch.alloc(&d_bbox, 6));
    if (R >= 1 && R <= kMbMaxDepth && c->buildPath == 0) {
        // four launches whatever the depth: every level ranked and emitted at once (Morton order == BFS order within a level)
        const int rc = build_octree_morton(c, s, scratch, R, dimX, dimY, dimZ, voxels, e0, e1, &total, &internal, box);
        if (rc != RTO_OK) return rc;
        haveBox = true;
    } else {
        RTO_HIP(c, hipEventRecord(e0, s));
        RTO_HIP(c, hipMemcpyAsync(d_vox, voxels, nvox, hipMemcpyHostToDevice, s));
        RTO_HIP(c, hipEventRecord(e1, s));
        // level-by-level form (any depth up to kMaxDepth; also the cross-check of the other: rto_debug_set_build_path)
        // ---- occupancy pyramid, bottom-up; every level also leaves its number of mixed cells = internal nodes
        PyramidView V;
        std::memset(&V, 0, sizeof V);
        V.level[0] = d_vox; V.nx[0] = dimX; V.ny[0] = dimY; V.nz[0] = dimZ;
        LevelCountView LC;
        std::memset(&LC, 0, sizeof LC);
        for (int l = 1; l <= R; l++) {
            V.nx[l] = (V.nx[l - 1] + 1) / 2; V.ny[l] = (V.ny[l - 1] + 1) / 2; V.nz[l] = (V.nz[l - 1] + 1) / 2;
            const size_t cells = (size_t)V.nx[l] * V.ny[l] * V.nz[l];
            const bool wide = l == 1 && dimX % 16 == 0;         // level 1 reads the voxels 16 B at a time when rows stay aligned
            const unsigned nbl = (unsigned)(((wide ? cells / 8 : cells) + kBlock - 1) / kBlock);
            uint8_t* d = nullptr;
            int* d_bm = nullptr;
            RTO_HIP(c, scratch.alloc(&d, cells));
            RTO_HIP(c, scratch.alloc(&d_bm, (size_t)nbl));
            V.level[l] = d;
            LC.blockMixed[l] = d_bm; LC.numBlocks[l] = (int)nbl;
            if (wide)
                hipLaunchKernelGGL(k_pyramid_level1_wide, dim3(nbl), dim3(kBlock), 0, s, d_vox, dimX, dimY, dimZ, d, V.nx[l], V.ny[l], V.nz[l], d_bm);
            else
                hipLaunchKernelGGL(k_pyramid_level, dim3(nbl), dim3(kBlock), 0, s,
                                   V.level[l - 1], V.nx[l - 1], V.ny[l - 1], V.nz[l - 1], d, V.nx[l], V.ny[l], V.nz[l], 1 << l, dimX, dimY, dimZ, d_bm);
        }
        RTO_HIP(c, hipGetLastError());
        std::vector<long long> mixed((size_t)R + 1, 0);
        if (R > 0) {
            long long* d_mixed = nullptr;
            RTO_HIP(c, scratch.alloc(&d_mixed, (size_t)R + 1));
            hipLaunchKernelGGL(k_sum_level_counts, dim3((unsigned)R), dim3(1024), 0, s, LC, d_mixed);
            RTO_HIP(c, hipGetLastError());
            RTO_HIP(c, hipMemcpyAsync(mixed.data() + 1, d_mixed + 1, (size_t)R * sizeof(long long), hipMemcpyDeviceToHost, s));
            RTO_HIP(c, hipStreamSynchronize(s));      // the only read-back before the tree is complete
        }

        // ---- pass 1: level-order node lists, internal flags and ranks.  Tree level L holds the cells of pyramid level
        //      R - L; its internal nodes are that level's mixed cells, so every size is known up front and the loop
        //      below runs without touching the host.
        std::vector<LevelBuf> levels;
        {
            int64_t m = 1;
            for (int L = 0; L <= R && m > 0; L++) {
                LevelBuf lb; lb.m = m;
                lb.k = (L < R) ? (int64_t)mixed[(size_t)(R - L)] : 0;
                if (lb.m > 0x7fffffff) return fail(c, RTO_E_UNSUPPORTED, "rto_build_octree: more than 2^31 nodes on one level");
                levels.push_back(lb);
                m = lb.k * 8;
            }
        }
        for (const LevelBuf& lb : levels) { total += lb.m; internal += lb.k; }
        if (total > 0x7fffffff) return fail(c, RTO_E_UNSUPPORTED, "rto_build_octree: node indices are int32 (GPUNodes.child)");
        // every allocation first (the result arrays are real hipMallocs, ~0.1 ms each), then all kernels back to back
        RTO_HIP(c, hipMalloc(&c->d_nodes, (size_t)total * sizeof(rto_node)));
        if (internal > 0) {
            RTO_HIP(c, hipMalloc(&c->d_desc, (size_t)internal * sizeof(uint2)));
            RTO_HIP(c, hipMalloc(&c->d_descFirstChild, (size_t)internal * sizeof(int)));
        }
        std::vector<int*> d_bc(levels.size(), nullptr), d_bb(levels.size(), nullptr);
        for (size_t L = 0; L < levels.size(); L++) {
            LevelBuf& lb = levels[L];
            const size_t nb = (size_t)((lb.m + kBlock - 1) / kBlock);
            RTO_HIP(c, scratch.alloc(&lb.coords, (size_t)lb.m));
            RTO_HIP(c, scratch.alloc(&lb.state, (size_t)lb.m)); RTO_HIP(c, scratch.alloc(&lb.flag, (size_t)lb.m));
            RTO_HIP(c, scratch.alloc(&lb.rank, (size_t)lb.m));
            RTO_HIP(c, scratch.alloc(&d_bc[L], nb)); RTO_HIP(c, scratch.alloc(&d_bb[L], nb));
        }
        RTO_HIP(c, hipMemsetAsync(levels[0].coords, 0, sizeof(int4), s));     // the root: (0, 0, 0)
        for (size_t L = 0; L < levels.size(); L++) {
            LevelBuf& lb = levels[L];
            const int lv = R - (int)L;
            const int nb = (int)((lb.m + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(k_build_classify, dim3(nb), dim3(kBlock), 0, s, V, lb.coords, lb.m, lv, lb.state, lb.flag, d_bc[L]);
            hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, s, d_bc[L], nb, d_bb[L], c->d_visibleCount);
            hipLaunchKernelGGL(k_build_rank_children, dim3(nb), dim3(kBlock), 0, s, lb.flag, lb.coords, lb.m, d_bb[L], (1 << lv) / 2, lb.rank,
                               L + 1 < levels.size() ? levels[L + 1].coords : (int4*)nullptr);
        }
        RTO_HIP(c, hipGetLastError());

        // ---- pass 2: node records + descriptors
        int64_t levelBase = 0, internalBase = 0;
        for (size_t L = 0; L < levels.size(); L++) {
            const LevelBuf& lb = levels[L];
            const bool hasNext = L + 1 < levels.size();
            const int nb = (int)((lb.m + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(k_build_emit, dim3(nb), dim3(kBlock), 0, s, lb.coords, lb.state, lb.rank, lb.m, 1 << (R - (int)L),
                               levelBase, internalBase, hasNext ? levels[L + 1].state : nullptr, hasNext ? levels[L + 1].rank : nullptr,
                               internalBase + lb.k, c->d_nodes, c->d_desc, c->d_descFirstChild);
            levelBase += lb.m; internalBase += lb.k;
        }
        RTO_HIP(c, hipGetLastError());
    }
    // ---- where the solid geometry is (launch order, screen rectangle of the frames)
    if (!haveBox) {
        const int initBox[6] = { 0x7fffffff, 0x7fffffff, 0x7fffffff, -0x7fffffff, -0x7fffffff, -0x7fffffff };
        RTO_HIP(c, hipMemcpyAsync(d_bbox, initBox, sizeof initBox, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_solid_bbox, dim3((unsigned)std::min<int64_t>((total + kBlock - 1) / kBlock, 512)), dim3(kBlock), 0, s, c->d_nodes, total, d_bbox);
        RTO_HIP(c, hipGetLastError());
        RTO_HIP(c, hipEventRecord(e2, s));
        RTO_HIP(c, hipMemcpyAsync(box, d_bbox, sizeof box, hipMemcpyDeviceToHost, s));
    } else RTO_HIP(c, hipEventRecord(e2, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    for (int a = 0; a < 3; a++) {
        c->solidCentre[a] = box[a] <= box[3 + a] ? 0.5f * (float)(box[a] + box[3 + a]) : 0.5f * (float)(1 << R);
        c->solidLo[a] = box[a] <= box[3 + a] ? box[a] : 1; c->solidHi[a] = box[a] <= box[3 + a] ? box[3 + a] : 0;
    }
    RTO_HIP(c, hipEventElapsedTime(&c->buildUploadMs, e0, e1));
    RTO_HIP(c, hipEventElapsedTime(&c->buildMs, e1, e2));

    c->numNodes = total; c->visibleNodes = total; c->numInternal = internal;
    c->rootSize = 1 << R; c->depth = R;
    c->canonical = internal > 0;                 // a one-node tree is rendered by the generic kernel
    return build_cells(c);
}

int rto_debug_set_build_path(rto_context* c, int level_by_level) {
    if (!c) return RTO_E_INVALID;
    c->buildPath = level_by_level ? 1 : 0;
    return RTO_OK;
}

int rto_last_build_ms(const rto_context* c, float* kernels_ms, float* upload_ms) {
    if (!c || c->buildMs < 0.f) return RTO_E_INVALID;
    if (kernels_ms) *kernels_ms = c->buildMs;
    if (upload_ms) *upload_ms = c->buildUploadMs;
    return RTO_OK;
}

int rto_download_nodes(rto_context* c, rto_node* out, int64_t capacity, int64_t* count) {
    if (!c || !count) return RTO_E_INVALID;
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_download_nodes: no octree resident");
    *count = c->numNodes;
    if (!out) return RTO_OK;
    if (capacity < c->numNodes) return fail(c, RTO_E_INVALID, "rto_download_nodes: capacity too small");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    RTO_HIP(c, hipMemcpy(out, c->d_nodes, (size_t)c->numNodes * sizeof(rto_node), hipMemcpyDeviceToHost));
    return RTO_OK;
}

int rto_debug_tile_cost(rto_context* c, int32_t* host_cost, int64_t capacity, int64_t* count) {
    if (!c || !count) return RTO_E_INVALID;
    auto it = c->orders.find(c->lastOrderKey);
    *count = it == c->orders.end() ? 0 : it->second.tiles;
    if (!host_cost) return RTO_OK;
    if (it == c->orders.end() || capacity < it->second.tiles || !it->second.d_tileCost)
        return fail(c, RTO_E_INVALID, "rto_debug_tile_cost: nothing recorded / capacity");
    RTO_HIP(c, hipDeviceSynchronize());
    RTO_HIP(c, hipMemcpy(host_cost, it->second.d_tileCost, (size_t)it->second.tiles * sizeof(int), hipMemcpyDeviceToHost));
    return RTO_OK;
}

int rto_debug_set_tile_order(rto_context* c, const int32_t* host_order, int64_t n) {
    if (!c) return RTO_E_INVALID;
    auto it = c->orders.find(c->lastOrderKey);
    if (!host_order) {
        if (it == c->orders.end()) return RTO_OK;
        rto_context::OrderState& o = it->second;
        o.fixed = false; o.tab[0].valid = false;
        return RTO_OK;
    }
    if (it == c->orders.end() || n != it->second.tiles || !it->second.tab[0].d)
        return fail(c, RTO_E_INVALID, "rto_debug_set_tile_order: render one frame first; n must equal the tile count");
    const int tilesX = (int)((it->second.key[0] + 7) / 8);                 // key[0] = W of the frames the tables belong to
    if (tilesX <= 0 || tilesX > 0xffff || n / tilesX > 0xffff) return fail(c, RTO_E_INVALID, "rto_debug_set_tile_order: frame too large for packed entries");
    std::vector<int> packed((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (host_order[i] < 0 || host_order[i] >= n) return fail(c, RTO_E_INVALID, "rto_debug_set_tile_order: entry out of range");
        packed[(size_t)i] = (host_order[i] % tilesX) | ((host_order[i] / tilesX) << 16);      // the kernels' slot table holds tx | ty << 16
    }
    RTO_HIP(c, hipDeviceSynchronize());
    RTO_HIP(c, hipMemcpy(it->second.tab[0].d, packed.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    it->second.fixed = true; it->second.tab[0].valid = true;
    return RTO_OK;
}

int rto_set_launch_order(rto_context* c, int policy, int refresh_period) {
    if (!c) return RTO_E_INVALID;
    if (policy != RTO_ORDER_CENTRE_OUT && policy != RTO_ORDER_TEMPORAL) return fail(c, RTO_E_INVALID, "rto_set_launch_order: unknown policy");
    if (refresh_period < 0) return fail(c, RTO_E_INVALID, "rto_set_launch_order: refresh_period must be >= 0 (0 keeps the current one)");
    c->orderPolicy = policy;
    if (refresh_period > 0) c->orderPeriod = refresh_period;
    for (auto& kv : c->orders) { if (!kv.second.fixed) kv.second.tab[0].valid = false; kv.second.tab[1].valid = false; kv.second.tab[0].age = kv.second.tab[1].age = 0; kv.second.tab[0].stretch = kv.second.tab[1].stretch = 1; }
    return RTO_OK;
}

int rto_forget_stream(rto_context* c, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    bool synced = false;
    for (auto it = c->orders.begin(); it != c->orders.end();) {                  // every kind of frame that was launched on the stream
        if (it->first.first != (hipStream_t)hip_stream) { ++it; continue; }
        if (!synced) {
            RTO_HIP(c, hipSetDevice(c->device));
            RTO_HIP(c, hipDeviceSynchronize());  // no kernel still reads the tables
            synced = true;
        }
        (void)hipFree(it->second.d_tileCost); (void)hipFree(it->second.tab[0].d); (void)hipFree(it->second.tab[1].d); (void)hipFree(it->second.d_queue); (void)hipFree(it->second.d_tileMask);
        if (c->lastOrderKey == it->first) c->lastOrderKey = rto_context::OrderKey(nullptr, -1);
        it = c->orders.erase(it);
    }
    return RTO_OK;
}

int rto_debug_set_tile_mask(rto_context* c, int enabled) {
    if (!c) return RTO_E_INVALID;
    if (enabled < 0 || enabled > 2) return fail(c, RTO_E_INVALID, "rto_debug_set_tile_mask: 0 off, 1 on, 2 on and complete before the frame starts");
    c->maskMode = enabled;
    return RTO_OK;
}

int rto_debug_tile_mask_info(const rto_context* c, int* level, int* num_cells) {
    if (!c) return RTO_E_INVALID;
    if (level) *level = c->cellLevel;
    if (num_cells) *num_cells = c->numCells;
    return RTO_OK;
}

// Developer aid (not in rto_hip.h): the three 64-bit frame counters as they stand; zero != 0 clears them first.  Used by
// tools/tri_profile.py with an A/B build (-DRTO_TRI_PROFILE) whose colour frames count their loop trips there.
int rto_debug_counters(rto_context* c, unsigned long long out[3], int zero) {
    if (!c || !out) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    if (zero) RTO_HIP(c, hipMemset(c->d_counters, 0, sizeof(Counters)));
    RTO_HIP(c, hipMemcpy(out, c->d_counters, sizeof(Counters), hipMemcpyDeviceToHost));
    return RTO_OK;
}

// Developer aid (not in rto_hip.h): the first n ints of the instrumentation buffer -- the per-wave timeline records a
// -DRTO_TRI_PROFILE build's colour frames of the triangle path leave there (tools/tri_timeline.py).
int rto_debug_steps_buffer(rto_context* c, int32_t* out, int64_t n) {
    if (!c || !out || n <= 0) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    if (!c->d_steps || (size_t)n > c->stepsCap) return fail(c, RTO_E_INVALID, "rto_debug_steps_buffer: no such buffer");
    RTO_HIP(c, hipMemcpy(out, c->d_steps, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    return RTO_OK;
}

int rto_debug_sort_violations(rto_context* c, int* count) {
    if (!c || !count) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    RTO_HIP(c, hipMemcpy(count, c->d_sortViolations, sizeof(int), hipMemcpyDeviceToHost));
    return RTO_OK;
}

static int sync_cull_state(rto_context* c);

int rto_octree_info_get(const rto_context* c, rto_octree_info* out) {
    if (!c || !out) return RTO_E_INVALID;
    (void)sync_cull_state(const_cast<rto_context*>(c));   // visible_nodes of an asynchronous frustum update is fetched when asked for
    out->num_nodes = c->numNodes;
    out->num_internal = c->numInternal;
    out->root_size = c->rootSize;
    out->depth = c->depth;
    out->canonical = c->canonical ? 1 : 0;
    out->culling_active = c->culling ? 1 : 0;
    out->visible_nodes = c->visibleNodes;
    return RTO_OK;
}

int rto_set_kernel(rto_context* c, int kernel) {
    if (!c) return RTO_E_INVALID;
    if (kernel < RTO_KERNEL_AUTO || kernel > RTO_KERNEL_PACKED_V3)
        return fail(c, RTO_E_INVALID, "rto_set_kernel: unknown kernel id");
    if (kernel >= RTO_KERNEL_PACKED && c->numNodes > 0 && !c->canonical)
        return fail(c, RTO_E_UNSUPPORTED, "rto_set_kernel: packed kernel needs a canonical BFS octree");
    c->kernelMode = kernel;
    return RTO_OK;
}

// ---------------------------------------------------------------- frustum culling
// One frustum update for the given planes (LEFT, RIGHT, TOP, BOTTOM, NEAR, FAR; normalised) and margin.
static bool stream_is_capturing(hipStream_t s);

// The visibility state (descriptor bits, d_vis, d_start) was just rewritten by a kernel on c->stream.  Nothing is recorded
// here: an event record between the update and the frame that follows it on the same stream -- the common case, what
// RayTracerBVH::renderSceneComputeWithCulling does on one GPU -- cost 4.7 us per frame (measured: 0.0468 against 0.0421 ms per
// call).  The point is marked when somebody on another stream asks (order_after_cull).
static int cull_state_written(rto_context* c) {
    if (stream_is_capturing(c->stream)) return RTO_OK;      // the replays are launched by the caller, who orders them like any graph
    c->cullSeq++;
    return RTO_OK;
}

// A launch on stream `s` is about to READ the visibility state.  On the context's own stream the stream orders it; any other
// stream -- a caller's, or rto_comm's render stream: what RayTracerBVH::renderSceneComputeWithCulling uses on several GPUs --
// waits for the last rewrite: the first such launch after an update records an event on the context's stream (behind the
// update, and behind whatever else was queued there since), every such launch waits for it on its own stream
// (hipStreamWaitEvent: nothing happens on the host).  A stream that is being CAPTURED cannot wait for an event from outside
// its capture: there the host waits for the event instead.
static int order_after_cull(rto_context* c, hipStream_t s) {
    if (s == c->stream || c->cullSeq == 0) return RTO_OK;
    if (stream_is_capturing(c->stream)) return RTO_OK;      // an update inside a capture of c->stream: the caller's ordering
    if (c->evCullSeq != c->cullSeq) {
        RTO_HIP(c, hipEventRecord(c->evCull, c->stream));
        c->evCullSeq = c->cullSeq;
        c->evCullDone = false;
    }
    if (c->evCullDone) return RTO_OK;
    if (stream_is_capturing(s)) {
        // event queries count as "unsafe" calls under the global capture mode (what torch.cuda.graph uses): this thread steps
        // into the relaxed mode for the one call
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        RTO_HIP(c, hipThreadExchangeStreamCaptureMode(&mode));
        const hipError_t e = hipEventSynchronize(c->evCull);
        (void)hipThreadExchangeStreamCaptureMode(&mode);
        if (e != hipSuccess) return fail(c, RTO_E_HIP, std::string("order_after_cull: ") + hipGetErrorString(e));
        c->evCullDone = true;
        return RTO_OK;
    }
    RTO_HIP(c, hipStreamWaitEvent(s, c->evCull, 0));
    return RTO_OK;
}

// What the last (asynchronous) frustum update left on the device -> the host's copies (rootVisible, visibleNodes).  Only the
// paths that need them on the host call this: the A/B kernels, the generic kernel, compaction, info queries.
static int sync_cull_state(rto_context* c) {
    if (!c->culling || !c->cullAsync || !c->cullStateStale) return RTO_OK;
    if (stream_is_capturing(c->stream))
        return fail(c, RTO_E_UNSUPPORTED, "the result of a frustum update cannot be read while the context's stream is being captured");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    if (c->evCullSeq == c->cullSeq) c->evCullDone = true;
    StartState st;
    RTO_HIP(c, hipMemcpy(&st, c->d_start, sizeof st, hipMemcpyDeviceToHost));
    c->rootVisible = st.rootVisible ? 1 : 0;
    c->visibleNodes = st.visibleCount;
    if (st.visibleCount < 0) {
        // root visible: k_cull_desc left the flags and did not count them (its fast path) -- count now, on demand
        const int64_t n = c->numNodes;
        const int nb = (int)((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_vis_block_counts, dim3(nb), dim3(kBlock), 0, c->stream, c->d_vis, n, c->d_blockCount);
        hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, c->stream, c->d_blockCount, nb, c->d_blockBase, c->d_visibleCount);
        RTO_HIP(c, hipGetLastError());
        int64_t count = 0;
        RTO_HIP(c, hipMemcpyAsync(&count, c->d_visibleCount, sizeof count, hipMemcpyDeviceToHost, c->stream));
        RTO_HIP(c, hipStreamSynchronize(c->stream));
        c->visibleNodes = count;
    }
    if (!c->cullCaptured) c->cullStateStale = false;       // a captured update may be replayed at any time: always ask again
    return RTO_OK;
}

// remap + compaction of the current visibility flags into d_compact, on stream s
static int ensure_compact(rto_context* c, hipStream_t s) {
    if (!c->culling) return RTO_OK;
    if (stream_is_capturing(s))
        return fail(c, RTO_E_UNSUPPORTED, "render: the compacted node array of the last frustum update is made on first use; "
                                          "render one frame with this kernel before hipStreamBeginCapture");
    const int rcState = sync_cull_state(c);                // also orders this stream behind the update (device-wide wait)
    if (rcState != RTO_OK) return rcState;
    if (c->compactValid) return RTO_OK;
    const int64_t n = c->numNodes;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    if (c->cullAsync) {                                    // k_cull_desc leaves flags only: count and scan them now
        hipLaunchKernelGGL(k_vis_block_counts, dim3(nb), dim3(kBlock), 0, s, c->d_vis, n, c->d_blockCount);
        hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, s, c->d_blockCount, nb, c->d_blockBase, c->d_visibleCount);
    }
    hipLaunchKernelGGL(k_cull_remap, dim3(nb), dim3(kBlock), 0, s, c->d_vis, n, c->d_blockBase, c->d_remap);
    hipLaunchKernelGGL(k_cull_compact, dim3(nb), dim3(kBlock), 0, s, c->d_nodes, n, c->d_remap, c->d_compact);
    RTO_HIP(c, hipGetLastError());
    RTO_HIP(c, hipStreamSynchronize(s));          // later users may sit on other streams
    c->compactValid = true;
    return RTO_OK;
}

// Is every node of the tree visible under these planes, by a margin no float rounding can close?  Frustum::testAABB's
// positive-vertex test (node_visible) culls a node when (n . p) + d < 0 for one plane, p the corner of the node's box -- widened by
// `margin` -- farthest along n.  Every node's box lies inside the root's [g, g + rootSize vs]: per axis the p-vertex coordinate is at
// least g + margin where n > 0 and at most g + rootSize vs - margin where n < 0, so
//     L = sum_a (n_a > 0 ? n_a (g_a + margin) : n_a (g_a + rootSize vs - margin)) + d
// bounds every node's value from below (exact arithmetic).  The device evaluates the value with ~10 float operations on terms of
// magnitude M = sum_a |n_a| (|g_a| + rootSize vs + margin) + |d|: its rounding error is below 1e-6 M; L >= 1e-3 M for all six planes
// therefore proves that k_cull_desc would flag EVERY node visible.  With the reference's margin of 150 units (S/RT:755) that is
// the case for every camera near a scene a few units across (config 2: L ~ 148 against M ~ 260) -- the reference's CPU loop then
// passes every node too, every frame.
static bool every_node_visible(const rto_context* c, const float planes[24], float margin) {
    const double span = (double)c->rootSize * (double)c->voxelSize;
    if (!(margin >= 0.0f) || !std::isfinite(span)) return false;
    for (int i = 0; i < 6; i++) {
        const float* pl = planes + 4 * i;
        double L = (double)pl[3], M = std::fabs((double)pl[3]);
        for (int a = 0; a < 3; a++) {
            const double n = (double)pl[a], g = (double)c->gridMin[a];
            if (!std::isfinite(n) || !std::isfinite(g)) return false;
            L += n > 0.0 ? n * (g + (double)margin) : n * (g + span - (double)margin);
            M += std::fabs(n) * (std::fabs(g) + span + (double)margin);
        }
        if (!std::isfinite(L) || !std::isfinite(M) || !(L >= 1e-3 * M)) return false;
    }
    return true;
}

// Frames launched on OTHER streams since the last update read the descriptors / d_vis / d_start an update is about to rewrite:
// the device is waited for (always, once a frame was captured on another stream: its replays are not seen here).  Not while
// capturing.  Called only where something is going to be written.
static int wait_for_other_streams(rto_context* c) {
    if ((c->otherStreams || c->foreignCaptured) && !stream_is_capturing(c->stream)) { RTO_HIP(c, hipDeviceSynchronize()); c->otherStreams = false; }
    return RTO_OK;
}

static int update_frustum_planes(rto_context* c, const float planes[24], float margin) {
    const int64_t n = c->numNodes;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    const int nbInt = (int)((c->numInternal + kBlock - 1) / kBlock);
    const bool canon = c->canonical && nbInt > 0 && c->d_descPos;
    const bool capturing = stream_is_capturing(c->stream);
    // An update the host proves to be a no-op while the "every node visible" state already stands writes nothing: no hazard with
    // frames in flight on other streams (the multi-GPU path's render stream), so no device-wide wait either.
    if (c->d_vis && canon && c->cullShortcut && !capturing && !c->cullCaptured && c->visAllOnes && c->culling && every_node_visible(c, planes, margin)) {
        c->compactValid = false;
        c->lastUpdateProven = true;
        c->cullAsync = true; c->cullStateStale = false;
        c->rootVisible = 1; c->visibleNodes = n;
        return RTO_OK;
    }
    { const int rcW = wait_for_other_streams(c); if (rcW != RTO_OK) return rcW; }
    if (!c->d_vis) {
        if (capturing)
            return fail(c, RTO_E_UNSUPPORTED, "rto_update_frustum: the first update of an octree allocates its buffers; call it once before hipStreamBeginCapture");
        // all or nothing: a failed allocation leaves none of them behind (the next update starts over)
        hipError_t e = fallible_malloc(reinterpret_cast<void**>(&c->d_vis), (size_t)n);
        if (e == hipSuccess) e = fallible_malloc(reinterpret_cast<void**>(&c->d_remap), (size_t)n * sizeof(int));
        if (e == hipSuccess) e = fallible_malloc(reinterpret_cast<void**>(&c->d_blockCount), (size_t)nb * sizeof(int));
        if (e == hipSuccess) e = fallible_malloc(reinterpret_cast<void**>(&c->d_blockBase), (size_t)nb * sizeof(int));
        if (e == hipSuccess) e = fallible_malloc(reinterpret_cast<void**>(&c->d_compact), (size_t)n * sizeof(rto_node));
        if (canon) {
            if (e == hipSuccess) e = fallible_malloc(reinterpret_cast<void**>(&c->d_cullBlockCount), (size_t)nbInt * sizeof(int));
            if (e == hipSuccess) e = fallible_malloc(reinterpret_cast<void**>(&c->d_cullBlockFirst), (size_t)nbInt * sizeof(int));
        }
        if (e != hipSuccess) {
            free_cull_buffers(c);
            return fail(c, RTO_E_HIP, std::string("rto_update_frustum: buffer allocation: ") + hipGetErrorString(e));
        }
        if (canon) RTO_HIP(c, hipMemsetAsync(c->d_start, 0, sizeof(StartState), c->stream));      // the ticket starts at zero
    }
    CullParams C;
    std::memcpy(C.planes, planes, sizeof C.planes);
    std::memcpy(C.gridMin, c->gridMin, sizeof C.gridMin);
    C.voxelSize = c->voxelSize;
    C.margin = margin;
    // The compacted array itself (S/RT:765-802: remap + copy, 2 x 22 MB of traffic at config 2) is made when somebody asks for
    // it -- rto_download_visible_nodes, the generic kernel: ensure_compact().  The packed kernels render from the visibility
    // bits in the descriptors, which is all a frustum update has to refresh for them.
    c->compactValid = false;
    c->lastUpdateProven = false;
    if (canon && c->cullShortcut && !capturing && !c->cullCaptured && every_node_visible(c, planes, margin)) {
        c->lastUpdateProven = true;
        // Proven on the host: the update flags every node.  The device state that says so is written once; while it stands, an
        // update launches nothing at all.
        if (!c->visAllOnes) {
            RTO_HIP(c, hipMemsetAsync(c->d_vis, 1, (size_t)n, c->stream));
            hipLaunchKernelGGL(k_desc_visall, dim3((unsigned)((c->numInternal + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->numInternal, c->d_desc);
            hipLaunchKernelGGL(k_start_at_root, dim3(1), dim3(1), 0, c->stream, c->d_start, c->depth, (long long)n);
            RTO_HIP(c, hipGetLastError());
            c->visAllOnes = true;
            const int rcEv = cull_state_written(c);
            if (rcEv != RTO_OK) return rcEv;
        }
        c->culling = true; c->cullAsync = true; c->cullStateStale = false;
        c->rootVisible = 1; c->visibleNodes = n;
        return RTO_OK;
    }
    c->visAllOnes = false;
    if (canon) {
        // one launch, nothing read back, capturable: flags, descriptor masks, and -- on the device -- where traversals start
        hipLaunchKernelGGL(k_cull_desc, dim3(nbInt), dim3(kBlock), 0, c->stream, C, c->d_descPos, c->d_descFirstChild, c->d_nodes, c->numInternal, c->depth,
                           c->d_desc, c->d_vis, c->d_cullBlockCount, c->d_cullBlockFirst, c->d_start);
        RTO_HIP(c, hipGetLastError());
        c->culling = true; c->cullAsync = true; c->cullStateStale = true;
        if (capturing) c->cullCaptured = true;
        return cull_state_written(c);
    }
    // arbitrary arrays (generic kernel): per-node flags + scan, the count and the root's flag come back in one copy
    if (capturing) return fail(c, RTO_E_UNSUPPORTED, "rto_update_frustum: capturable for canonical BFS octrees only");
    hipLaunchKernelGGL(k_cull_flags, dim3(nb), dim3(kBlock), 0, c->stream, C, c->d_nodes, n, c->d_vis, c->d_blockCount, c->d_visibleCount + 1);
    hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, c->stream, c->d_blockCount, nb, c->d_blockBase, c->d_visibleCount);
    RTO_HIP(c, hipGetLastError());
    int64_t back[2] = { 0, 0 };                     // visible nodes, the root's flag: one read-back
    RTO_HIP(c, hipMemcpyAsync(back, c->d_visibleCount, sizeof back, hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    c->visibleNodes = back[0];
    c->rootVisible = back[1] ? 1 : 0;
    c->culling = true; c->cullAsync = false; c->cullStateStale = false;
    return RTO_OK;
}

int rto_update_frustum(rto_context* c, const float view[16], float fov_deg, float aspect, int enable) {
    if (!c) return RTO_E_INVALID;
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_update_frustum: no octree uploaded");
    RTO_HIP(c, hipSetDevice(c->device));
    // The visibility masks live in the descriptors every traversal kernel reads.  Frames on the context's own stream
    // (rto_render_resident: what RayTracerBVH uses) are ordered around the update by the stream itself: nothing is waited for.
    // Frames launched on OTHER streams since the last update must be done before the masks change (wait_for_other_streams) --
    // but only where the update writes: one the host proves to change nothing returns without touching the device.
    if (!enable) {
        { const int rcW = wait_for_other_streams(c); if (rcW != RTO_OK) return rcW; }
        const int nbInt = (int)((c->numInternal + kBlock - 1) / kBlock);
        if (c->culling && c->canonical && nbInt > 0) {
            hipLaunchKernelGGL(k_desc_visall, dim3(nbInt), dim3(kBlock), 0, c->stream, c->numInternal, c->d_desc);
            RTO_HIP(c, hipGetLastError());
            const int rcEv = cull_state_written(c);
            if (rcEv != RTO_OK) return rcEv;
        }
        c->culling = false; c->cullAsync = false; c->cullStateStale = false; c->rootVisible = 1; c->visibleNodes = c->numNodes;
        c->visAllOnes = false;              // d_start / d_vis are not maintained while culling is off
        return RTO_OK;
    }
    if (!view) return fail(c, RTO_E_INVALID, "rto_update_frustum: view is NULL");
    // S/RT:731-734: Frustum(perspective(radians(fov), aspect, 0.01, 5000) * view), margin 150 (S/RT:755)
    float planes[24];
    rtmath::mat4 proj = rtmath::perspective(rtmath::radians(fov_deg), aspect, 0.01f, 5000.f);
    rtmath::mat4 vp = proj * rtmath::mat4::from(view);
    rtmath::frustum_planes(vp, planes);
    return update_frustum_planes(c, planes, 150.0f);
}

// Test / A-B hook (not in rto_hip.h): 0 = the general child test even on a grid whose planes are exact; 1 = automatic (default).
// Takes effect at once (the flag travels with every frame's parameters); info[0] = the grid is exact, info[1] = the 9-plane form is in use.
int rto_debug_set_exact_grid(rto_context* c, int enabled, int info[2]) {
    if (!c) return RTO_E_INVALID;
    c->exactGridAllowed = enabled != 0;
    const bool exact = c->numNodes > 0 && grid_is_exact(c->gridMin, c->voxelSize, c->rootSize);
    c->exactGrid = c->exactGridAllowed && exact;
    if (info) { info[0] = exact ? 1 : 0; info[1] = c->exactGrid ? 1 : 0; }
    return RTO_OK;
}

int rto_debug_set_frustum_shortcut(rto_context* c, int enabled) {
    if (!c) return RTO_E_INVALID;
    c->cullShortcut = enabled != 0;
    if (!c->cullShortcut) c->visAllOnes = false;
    return RTO_OK;
}

int rto_debug_last_frustum_update_proven(const rto_context* c, int* proven) {
    if (!c || !proven) return RTO_E_INVALID;
    *proven = c->lastUpdateProven ? 1 : 0;
    return RTO_OK;
}

int rto_debug_update_frustum_planes(rto_context* c, const float planes[24], float margin) {
    if (!c) return RTO_E_INVALID;
    if (!planes) return fail(c, RTO_E_INVALID, "rto_debug_update_frustum_planes: planes is NULL");
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_debug_update_frustum_planes: no octree uploaded");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    return update_frustum_planes(c, planes, margin);
}

int rto_download_visible_nodes(rto_context* c, rto_node* out, int64_t capacity, int64_t* count) {
    if (!c || !count) return RTO_E_INVALID;
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_download_visible_nodes: no octree uploaded");
    RTO_HIP(c, hipSetDevice(c->device));
    const int rcState = sync_cull_state(c);
    if (rcState != RTO_OK) return rcState;
    *count = c->visibleNodes;
    if (!out) return RTO_OK;
    if (capacity < c->visibleNodes) return fail(c, RTO_E_INVALID, "rto_download_visible_nodes: capacity too small");
    const int rcCompact = ensure_compact(c, c->stream);
    if (rcCompact != RTO_OK) return rcCompact;
    const rto_node* src = c->culling ? c->d_compact : c->d_nodes;
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    RTO_HIP(c, hipMemcpy(out, src, (size_t)c->visibleNodes * sizeof(rto_node), hipMemcpyDeviceToHost));
    return RTO_OK;
}

// ---------------------------------------------------------------- render
int rto_partition_rows(const rto_frame* f, const rto_partition* p) {
    if (!f || f->height <= 0) return 0;
    if (!p || p->num_parts <= 1) return f->height;
    if (p->band_rows <= 0 || p->part < 0 || p->part >= p->num_parts) return 0;
    const int bands = (f->height + p->band_rows - 1) / p->band_rows;
    int rows = 0;
    for (int b = p->part; b < bands; b += p->num_parts) {
        int lo = b * p->band_rows, hi = lo + p->band_rows;
        rows += (hi < f->height ? hi : f->height) - lo;
    }
    return rows;
}

}  // extern "C"

static bool stream_is_capturing(hipStream_t s);

// Conservative pixel rectangle (inclusive) of the world-space box [lo, hi]: rays through pixels outside it miss the box
// for certain.  Valid only when all 8 corners are strictly in front of the eye (then the box projects onto the convex
// hull of its projected corners); computed in double with a margin of 2 pixels plus the box's float rounding, far above
// any float error of the per-pixel ray directions.  Otherwise: the whole image.  x0 > x1 or y0 > y1: nothing can hit.
static void screen_rectangle(const rto_frame* f, float tanHalfFov, const double lo[3], const double hi[3], int out[4]) {
    struct { int W, H; } P{ f->width, f->height };
    out[0] = 0; out[1] = 0; out[2] = P.W - 1; out[3] = P.H - 1;
    const double tanH = (double)tanHalfFov, asp = (double)f->aspect;
    double lox = 1e300, loy = 1e300, hix = -1e300, hiy = -1e300;
    bool allInFront = tanH > 0.0 && asp > 0.0 && std::isfinite(hi[0] - lo[0]) && std::isfinite(hi[1] - lo[1]) && std::isfinite(hi[2] - lo[2]);
    // the kernels build the boxes in float (gridMin + float(c) * voxel): widen by a few float ulps of the largest coordinate
    double big = 0.0;
    for (int a = 0; a < 3; a++) big = std::max(big, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    const double pad = big * 4.0 * 1.1920929e-7;
    for (int k = 0; k < 8 && allInFront; k++) {
        const double wx = (k & 1) ? hi[0] + pad : lo[0] - pad, wy = (k & 2) ? hi[1] + pad : lo[1] - pad, wz = (k & 4) ? hi[2] + pad : lo[2] - pad;
        const float* v = f->view;
        const double vx = v[0] * wx + v[4] * wy + v[8] * wz + v[12], vy = v[1] * wx + v[5] * wy + v[9] * wz + v[13],
                     vz = v[2] * wx + v[6] * wy + v[10] * wz + v[14];
        if (!(vz < -1e-6 * (1.0 + std::fabs(vx) + std::fabs(vy)))) { allInFront = false; break; }
        const double sx = ((vx / -vz) / (asp * tanH) * 0.5 + 0.5) * P.W, sy = (0.5 - (vy / -vz) / tanH * 0.5) * P.H;
        if (!std::isfinite(sx) || !std::isfinite(sy)) { allInFront = false; break; }
        lox = std::min(lox, sx); hix = std::max(hix, sx); loy = std::min(loy, sy); hiy = std::max(hiy, sy);
    }
    if (allInFront) {
        const double x0 = std::floor(lox) - 2.0, x1 = std::ceil(hix) + 2.0, y0 = std::floor(loy) - 2.0, y1 = std::ceil(hiy) + 2.0;
        out[0] = (int)std::max(0.0, std::min(x0, (double)P.W));        // may exceed W-1: then nothing can hit
        out[2] = (int)std::max(-1.0, std::min(x1, (double)P.W - 1.0));
        out[1] = (int)std::max(0.0, std::min(y0, (double)P.H));
        out[3] = (int)std::max(-1.0, std::min(y1, (double)P.H - 1.0));
    }
}

// What the screen rectangles and the multi-GPU split plan need to know of a scene (rto_scene_bounds, include/rto_hip.h).
static rto_scene_bounds bounds_of(const rto_context* c) {
    rto_scene_bounds b;
    std::memcpy(b.grid_min, c->gridMin, sizeof b.grid_min);
    b.voxel_size = c->voxelSize;
    b.root_size = c->rootSize;
    for (int a = 0; a < 3; a++) { b.solid_lo[a] = c->solidLo[a]; b.solid_hi[a] = c->solidHi[a]; }
    return b;
}

// Pixel rectangles (inclusive) of the root box and of the solid leaves' bounding box for one frame: pure host arithmetic on
// the frame and the scene's bounds, so every rank of a screen split derives the same numbers (rto_split_plan_make).
static void frame_rectangles(const rto_scene_bounds& S, const rto_frame* f, float tanHalfFov, int root[4], int solid[4]) {
    const double ext = (double)S.root_size * (double)S.voxel_size;
    const double rlo[3] = { S.grid_min[0], S.grid_min[1], S.grid_min[2] };
    const double rhi[3] = { rlo[0] + ext, rlo[1] + ext, rlo[2] + ext };
    screen_rectangle(f, tanHalfFov, rlo, rhi, root);
    if (S.solid_lo[0] > S.solid_hi[0]) { solid[0] = solid[1] = 0; solid[2] = solid[3] = -1; return; }     // nothing solid: nothing to hit
    double slo[3], shi[3];
    for (int a = 0; a < 3; a++) {
        // widened by one voxel: the Marching-Cubes triangles of the surface cells (config 5) reach half a voxel beyond
        // the solid leaves; one rectangle serves both paths
        slo[a] = (double)S.grid_min[a] + (double)(S.solid_lo[a] - 1) * (double)S.voxel_size;
        shi[a] = (double)S.grid_min[a] + (double)(S.solid_hi[a] + 1) * (double)S.voxel_size;
    }
    int r[4];
    screen_rectangle(f, tanHalfFov, slo, shi, r);
    solid[0] = std::max(r[0], root[0]); solid[1] = std::max(r[1], root[1]);
    solid[2] = std::min(r[2], root[2]); solid[3] = std::min(r[3], root[3]);
}

// Launch geometry "one wave per tile of the whole image, every wave stores all its pixels" (generic / V1 / triangle kernels,
// instrumentation modes, caller-supplied orders).
static void whole_image_box(RenderParams& P) {
    P.boxX0 = 0; P.boxY0 = 0; P.boxW = P.tilesX; P.boxH = P.tilesY;
    P.traceWaves = P.launchWaves = P.tilesX * P.tilesY;
    P.skipOutside = 0; P.fillChunks = 0;
    P.fillTopRows = 0; P.fillBotRow0 = P.localRows; P.fillLeftW = 0; P.fillRightX0 = P.W;
    P.fillTopChunks = P.fillBotChunks = P.fillLeftPer = P.fillRightPer = 0;
}

// Launch geometry of the packed kernels' colour / shade frames: waves only for the tiles that can meet the root
// rectangle, the region outside it shared out as 64-pixel chunks (RenderParams, rto_device.hip.h).
static int local_row_of_first_global_at_least(const RenderParams& P, int gy) {
    // local rows are ordered by their global row: count the local rows whose global row is < gy
    if (gy <= 0) return 0;
    if (P.numParts == 1) return std::min(gy, P.localRows);
    int rows = 0;
    const int bands = (P.H + P.bandRows - 1) / P.bandRows;
    for (int b = P.part; b < bands; b += P.numParts) {
        const int lo = b * P.bandRows, hi = std::min(lo + P.bandRows, P.H);
        if (hi <= gy) rows += hi - lo;
        else { if (lo < gy) rows += gy - lo; break; }
    }
    return std::min(rows, P.localRows);
}

static void root_rectangle_box(RenderParams& P, int minFillWaves) {
    const bool nothing = !P.rootVisible || P.rootX0 > P.rootX1 || P.rootY0 > P.rootY1;   // no ray can meet the root box
    int lyA, lyB, x0, x1;                       // rows [lyA, lyB) and columns [x0, x1] hold the rectangle's pixels
    if (nothing) { lyA = lyB = P.localRows; x0 = 0; x1 = -1; }
    else {
        lyA = local_row_of_first_global_at_least(P, P.rootY0);
        lyB = local_row_of_first_global_at_least(P, P.rootY1 + 1);
        x0 = std::max(P.rootX0, 0); x1 = std::min(P.rootX1, P.W - 1);
        if (lyA >= lyB || x0 > x1) { lyA = lyB = P.localRows; x0 = 0; x1 = -1; }
    }
    if (lyA >= lyB) { P.boxX0 = P.boxY0 = 0; P.boxW = P.boxH = 0; }
    else {
        // tile box, rounded outwards to multiples of 4 tiles: under a moving camera the box (and with it the launch-order
        // table) then changes 4x less often; the extra tiles are waves without work
        static const int rnd = []() { const int v = (int)dev_env("RTO_BOX_ROUND", 0);      // A/B knob (dev builds)
                                      return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) ? v - 1 : 3; }();
        const int tx0 = (x0 / 8) & ~rnd, ty0 = (lyA / 8) & ~rnd;
        const int tx1 = std::min(P.tilesX - 1, ((x1 / 8) | rnd)), ty1 = std::min(P.tilesY - 1, (((lyB - 1) / 8) | rnd));
        P.boxX0 = tx0; P.boxY0 = ty0; P.boxW = tx1 - tx0 + 1; P.boxH = ty1 - ty0 + 1;
    }
    P.traceWaves = P.boxW * P.boxH;
    P.skipOutside = 1;
    P.fillTopRows = lyA; P.fillBotRow0 = lyB;
    P.fillLeftW = x0; P.fillRightX0 = x1 + 1;
    if (lyA >= lyB) { P.fillLeftW = 0; P.fillRightX0 = P.W; }           // no middle rows
    const long topPix = (long)lyA * P.W, botPix = (long)(P.localRows - lyB) * P.W;
    P.fillTopChunks = (int)((topPix + kWave - 1) / kWave);
    P.fillBotChunks = (int)((botPix + kWave - 1) / kWave);
    P.fillLeftPer = (P.fillLeftW + kWave - 1) / kWave;
    P.fillRightPer = (P.W - P.fillRightX0 + kWave - 1) / kWave;
    P.fillChunks = P.fillTopChunks + P.fillBotChunks + (lyB - lyA) * (P.fillLeftPer + P.fillRightPer);
    // a tiny (or empty) box still needs enough waves to write the outside region at memory speed
    P.launchWaves = std::max(P.traceWaves, std::min(P.fillChunks, minFillWaves));
    if (P.launchWaves <= 0) P.launchWaves = 1;
}

// launch_stream: the stream the frame is about to be launched on (to refuse work a capture in progress cannot hold)
static int fill_params(rto_context* c, const rto_frame* f, const rto_partition* p, RenderParams& P, hipStream_t launch_stream = nullptr) {
    if (!f) return fail(c, RTO_E_INVALID, "render: frame is NULL");
    if (f->width <= 0 || f->height <= 0) return fail(c, RTO_E_INVALID, "render: width/height must be positive");
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "render: no octree uploaded (setOctree first)");
    if (p && p->num_parts > 1 && (p->band_rows <= 0 || (p->band_rows % 8) != 0 || p->part < 0 || p->part >= p->num_parts))
        return fail(c, RTO_E_INVALID, "render: partition needs band_rows % 8 == 0 and 0 <= part < num_parts");
    rtmath::mat4 inv = rtmath::inverse(rtmath::mat4::from(f->view));        // S/RT:348
    std::memcpy(P.invView, inv.data(), sizeof P.invView);
    std::memcpy(P.camPos, f->cam_pos, sizeof P.camPos);
    P.aspect = f->aspect;
    P.tanHalfFov = std::tan(rtmath::radians(f->fov_deg) * 0.5f);               // S/RT:340,344
    std::memcpy(P.gridMin, c->gridMin, sizeof P.gridMin);
    P.voxelSize = c->voxelSize;
    rtmath::vec3 l = rtmath::normalize(rtmath::vec3(-1.0f, -1.0f, -1.0f));     // S/RT:333
    P.lightNeg[0] = -l.x; P.lightNeg[1] = -l.y; P.lightNeg[2] = -l.z;
    for (int a = 0; a < 3; a++) { const volatile float q = 1.0f / P.lightNeg[a]; P.lightInv[a] = q; }     // IEEE single division (volatile: never a reciprocal approximation)
    P.W = f->width; P.H = f->height;
    P.rootSize = c->rootSize; P.depth = c->depth > 0 ? c->depth : 1;
    P.exactGrid = c->exactGrid ? 1 : 0;
    P.numParts = (p && p->num_parts > 1) ? p->num_parts : 1;
    P.part = P.numParts > 1 ? p->part : 0;
    P.bandRows = P.numParts > 1 ? p->band_rows : f->height;
    P.localRows = rto_partition_rows(f, p);
    P.tilesX = (P.W + 7) / 8;
    P.tilesY = (P.localRows + 7) / 8;
    P.rootVisible = c->rootVisible;
    // S/RT:341-346: nx depends on the column only, ny on the row only -> two small tables, same float ops
    if (c->rayW != P.W || c->rayH != P.H || c->rayAspect != P.aspect || c->rayTan != P.tanHalfFov || !c->d_rayX) {
        if (launch_stream && stream_is_capturing(launch_stream))
            return fail(c, RTO_E_UNSUPPORTED, "render: a frame of a new width/height/aspect/fov rebuilds the ray tables (allocation + "
                                              "synchronisation); render one such frame before hipStreamBeginCapture");
        std::vector<float> tx((size_t)P.W), ty((size_t)P.H);
        for (int px = 0; px < P.W; px++) {
            float nx = ((float)px + 0.5f) / (float)P.W * 2.0f - 1.0f;
            nx *= P.aspect;
            nx *= P.tanHalfFov;
            tx[(size_t)px] = nx;
        }
        for (int py = 0; py < P.H; py++) {
            float ny = 1.0f - ((float)py + 0.5f) / (float)P.H * 2.0f;
            ny *= P.tanHalfFov;
            ty[(size_t)py] = ny;
        }
        if (c->rayW != P.W || !c->d_rayX) { (void)hipFree(c->d_rayX); c->d_rayX = nullptr; RTO_HIP(c, hipMalloc(&c->d_rayX, tx.size() * sizeof(float))); }
        if (c->rayH != P.H || !c->d_rayY) { (void)hipFree(c->d_rayY); c->d_rayY = nullptr; RTO_HIP(c, hipMalloc(&c->d_rayY, ty.size() * sizeof(float))); }
        RTO_HIP(c, hipDeviceSynchronize());   // nothing may still read the old tables
        RTO_HIP(c, hipMemcpy(c->d_rayX, tx.data(), tx.size() * sizeof(float), hipMemcpyHostToDevice));
        RTO_HIP(c, hipMemcpy(c->d_rayY, ty.data(), ty.size() * sizeof(float), hipMemcpyHostToDevice));
        c->rayW = P.W; c->rayH = P.H; c->rayAspect = P.aspect; c->rayTan = P.tanHalfFov;
    }
    P.rayX = c->d_rayX; P.rayY = c->d_rayY;
    P.tileOrder = nullptr; P.tileCost = nullptr;
    P.start = nullptr;                     // the lean kernels' launchers point it at the device-side start state while culling is active
    P.tileMask = nullptr; P.maskStamp = 0; P.maskAllIndex = 0; P.maskBlocks = 0; P.maskTrustSlots = 0; P.maskCells = nullptr; P.maskNumCells = 0; P.maskLdsBytes = 0;
    P.maskInvAspTan = P.maskInvTanH = 0.0f;                                 // prepare_schedule switches the occupancy mask on for lean colour / shade frames
    for (int r = 0; r < 3; r++) for (int k = 0; k < 4; k++) P.viewRows[r * 4 + k] = f->view[k * 4 + r];   // column-major glm matrix -> rows
    {   // rays through pixels outside these rectangles miss the root box / every solid leaf for certain
        int rr[4], sr[4];
        frame_rectangles(bounds_of(c), f, P.tanHalfFov, rr, sr);
        P.rootX0 = rr[0]; P.rootY0 = rr[1]; P.rootX1 = rr[2]; P.rootY1 = rr[3];
        P.solidX0 = sr[0]; P.solidY0 = sr[1]; P.solidX1 = sr[2]; P.solidY1 = sr[3];
    }
    {   // project the centre of the solid geometry; any value is valid, it only orders the launch
        const rtmath::mat4 V = rtmath::mat4::from(f->view);
        const float wc[3] = { c->gridMin[0] + c->solidCentre[0] * c->voxelSize, c->gridMin[1] + c->solidCentre[1] * c->voxelSize,
                              c->gridMin[2] + c->solidCentre[2] * c->voxelSize };
        const float vx = V[0][0] * wc[0] + V[1][0] * wc[1] + V[2][0] * wc[2] + V[3][0];
        const float vy = V[0][1] * wc[0] + V[1][1] * wc[1] + V[2][1] * wc[2] + V[3][1];
        const float vz = V[0][2] * wc[0] + V[1][2] * wc[1] + V[2][2] * wc[2] + V[3][2];
        float sx = 0.5f * (float)P.W, sy = 0.5f * (float)P.H;
        if (vz < 0.0f && P.tanHalfFov > 0.0f && P.aspect > 0.0f) {
            const float ndcx = vx / (-vz) / (P.aspect * P.tanHalfFov), ndcy = vy / (-vz) / P.tanHalfFov;
            if (std::isfinite(ndcx) && std::isfinite(ndcy)) { sx = (ndcx * 0.5f + 0.5f) * (float)P.W; sy = (0.5f - ndcy * 0.5f) * (float)P.H; }
        }
        int tcx = (int)std::floor(sx / 8.0f), tcy = (int)std::floor(sy / 8.0f / (float)P.numParts);
        P.orderCx = tcx < 0 ? 0 : (tcx >= P.tilesX ? P.tilesX - 1 : tcx);
        P.orderCy = tcy < 0 ? 0 : (tcy >= P.tilesY ? P.tilesY - 1 : tcy);
        if (P.tilesX <= 0) P.orderCx = 0;
        if (P.tilesY <= 0) P.orderCy = 0;
    }
    whole_image_box(P);
    return RTO_OK;
}

// Is `s` being captured into a HIP graph?  Then the launch must stay free of anything a replay could not repeat:
// no timing events (events recorded inside a capture cannot be timed), no launch-order rebuild (see launch_trace).
static bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
}

// The scheduling state of launch stream `s` (temporal launch order, persistent-kernel counter).  At most
// kMaxOrderStreams streams are tracked; when a new stream arrives at the limit, the entry that was used longest ago is
// dropped (its buffers are freed after a device synchronise, so a capture in progress refuses instead: see callers).
static rto_context::OrderState* order_state(rto_context* c, hipStream_t s, int path, bool capturing) {
    const rto_context::OrderKey key(s, path);
    auto it = c->orders.find(key);
    if (it != c->orders.end()) { it->second.lastUse = ++c->orderClock; return &it->second; }
    if (c->orders.size() >= rto_context::kMaxOrderStreams) {
        if (capturing) return nullptr;                       // eviction frees memory: not inside a capture
        auto victim = c->orders.begin();
        for (auto jt = c->orders.begin(); jt != c->orders.end(); ++jt)
            if (jt->second.lastUse < victim->second.lastUse) victim = jt;
        (void)hipDeviceSynchronize();
        (void)hipFree(victim->second.d_tileCost); (void)hipFree(victim->second.tab[0].d); (void)hipFree(victim->second.tab[1].d);
        (void)hipFree(victim->second.d_queue); (void)hipFree(victim->second.d_tileMask);
        if (c->lastOrderKey == victim->first) c->lastOrderKey = rto_context::OrderKey(nullptr, -1);
        c->orders.erase(victim);
    }
    rto_context::OrderState* st = &c->orders[key];
    st->lastUse = ++c->orderClock;
    return st;
}

// The occupancy mask serves frames cut into at most this many parts: a part's launch projects every cell for its share of the
// rows, which pays as long as the share is not too small -- a rank of the screen split, per frame: 2 GPUs 28.1 -> 25.6 us, 4 GPUs
// 21.9 -> 20.2, 8 GPUs (7 rendering parts) 14.4 -> 14.3.  The triangle frames (path 1) keep it for whole frames: a rank of 8 at
// config 5 measured 93.5 us with it against 91.5 without.  RTO_MASK_PARTS_MAX overrides both (A/B runs; 1: whole frames only).
static int mask_parts_max(int path) {
    static const int v = (int)dev_env("RTO_MASK_PARTS_MAX", 0);
    return v > 0 ? v : (path == 1 ? 1 : 8);
}

// Launch geometry + launch order of one frame of the packed kernels on stream `s` (shared by the octree and the triangle
// path): waves for the tiles of `rect`'s box only, the outside shared out as fill chunks, the box's tiles in the order of
// the costs earlier frames recorded.  frameMode: a colour / shade frame (records costs); otherwise (instrumentation) the
// whole image keeps one wave per tile.  On return Q holds the geometry, the table and the cost pointer.
constexpr int kMovedPeriod = 4;     // a camera in motion: the launch-order table is rebuilt every 4th frame (see prepare_schedule)
static int prepare_schedule(rto_context* c, hipStream_t s, bool capturing, bool frameMode, bool timelineMode, int path, const int rect[4],
                            RenderParams& Q, rto_context::OrderState** stOut, int maskRegion = -1, bool single = false) {
    const RenderParams& P = Q;
    const int tiles = P.tilesX * P.tilesY;
    Q.tileOrder = nullptr; Q.tileCost = nullptr;
    const long key[7] = { P.W, P.H, P.numParts, P.part, P.bandRows, tiles, path };
    rto_context::OrderState* st = order_state(c, s, path, capturing);      // the scheduling state of this kind of frame on this stream
    *stOut = st;
    // occupancy mask (maskRegion >= 0: the caller launches a lean kernel; its first maskBlocks workgroups build the mask: mask_block)
    Q.tileMask = nullptr; Q.maskBlocks = 0;
    // (frames in up to mask_parts_max(path) parts: the ranks of the multi-GPU split included)
    if (st && maskRegion >= 0 && (frameMode || timelineMode) && c->maskMode != 0 && c->numCells > 0 && P.aspect > 0.0f && P.tanHalfFov > 0.0f &&
        (P.numParts <= mask_parts_max(path) || c->maskMode == 2)) {
        const int strips = (P.H + 7) / 8;
        const size_t words = (size_t)strips * P.tilesX + 3;              // tiles, "whole frame", "complete", ticket
        if (st->maskWords != words) {
            if (capturing)
                return fail(c, RTO_E_UNSUPPORTED, "render: the first frame of a new size on a stream allocates its occupancy mask; "
                                                  "render one such frame before hipStreamBeginCapture");
            (void)hipFree(st->d_tileMask); st->d_tileMask = nullptr; st->maskWords = 0;      // hipFree waits for the device
            RTO_HIP(c, hipMalloc(&st->d_tileMask, words * kMaxBatch * sizeof(unsigned)));
            RTO_HIP(c, hipMemset(st->d_tileMask, 0, words * kMaxBatch * sizeof(unsigned)));   // stamp 0 is never issued; tickets start at 0
            st->maskWords = words;
        }
        if (++c->maskStamp == 0) ++c->maskStamp;
        Q.tileMask = st->d_tileMask + (size_t)maskRegion * words;
        Q.maskStamp = c->maskStamp;
        Q.maskAllIndex = (int)(words - 3);
        Q.maskBlocks = (c->numCells + lean_block(path) - 1) / lean_block(path);
        static const int trustEnv = (int)dev_env("RTO_MASK_TRUST", -1);      // A/B knob (dev builds)
        Q.maskTrustSlots = c->maskMode == 2 ? 0 : (trustEnv >= 0 ? trustEnv : kMaskTrustSlots);
        Q.maskCells = c->d_cells; Q.maskNumCells = c->numCells;
        Q.maskInvAspTan = 1.0f / (P.aspect * P.tanHalfFov); Q.maskInvTanH = 1.0f / P.tanHalfFov;
    }
    rto_context::OrderState* o = (c->orderPolicy == RTO_ORDER_TEMPORAL && (frameMode || timelineMode) &&
                                  P.tilesX <= 0xffff && P.tilesY <= 0x7fff) ? st : nullptr;      // table entries are x | y << 16 inside the box
    if (o) {
        c->lastOrderKey = rto_context::OrderKey(s, path);
        if (o->tiles != tiles) {
            if (capturing)
                return fail(c, RTO_E_UNSUPPORTED, "render: the first frame of a new size on a stream allocates its launch-order "
                                                  "tables; render one such frame before hipStreamBeginCapture");
            // hipFree waits for the device: no kernel still reads the old tables
            (void)hipFree(o->d_tileCost); (void)hipFree(o->tab[0].d); (void)hipFree(o->tab[1].d);
            o->d_tileCost = o->tab[0].d = o->tab[1].d = nullptr; o->tiles = 0;
            RTO_HIP(c, hipMalloc(&o->d_tileCost, (size_t)tiles * sizeof(int)));
            RTO_HIP(c, hipMalloc(&o->tab[0].d, (size_t)tiles * sizeof(int)));
            RTO_HIP(c, hipMalloc(&o->tab[1].d, (size_t)tiles * sizeof(int)));
            RTO_HIP(c, hipMemset(o->d_tileCost, 0, (size_t)tiles * sizeof(int)));
            o->tiles = tiles; o->tab[0].valid = o->tab[1].valid = false; o->costValid = false; o->fixed = false;
        }
        if (std::memcmp(key, o->key, sizeof key) != 0) { o->tab[0].valid = o->tab[1].valid = false; o->costValid = false; o->fixed = false; std::memcpy(o->key, key, sizeof key); }
        o->active = capturing ? 1 : 0;
        if (capturing) {
            // a capture's first frame on this stream always gets a rebuild node of its own: whatever plain launches, another
            // graph or this graph's own tail leave in the table, a replay starts from a table built for ITS first frame's box
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            unsigned long long id = 0;
            if (hipStreamGetCaptureInfo(s, &cs, &id) != hipSuccess) id = ~0ull;
            if (id != o->capId || id == ~0ull) { o->tab[1].valid = false; o->capId = id; }
        }
    }
    const bool fixedOrder = o && o->fixed && !capturing;       // the caller's table serves plain launches only
    // launch geometry: frames get waves for the rectangle's tiles only and share the outside region out as wide stores;
    // instrumentation modes and caller-supplied orders keep one wave per tile of the image
    if ((frameMode || timelineMode) && !fixedOrder) {
        // a frame only needs the pixels that can meet SOLID geometry: every other ray is black whatever it pops on the way
        // (S/RT:363), so the rectangle of the solid leaves' bounding box replaces the root box's
        Q.rootX0 = rect[0]; Q.rootY0 = rect[1]; Q.rootX1 = rect[2]; Q.rootY1 = rect[3];
        root_rectangle_box(Q, 8 * c->numCUs);
        static const bool trace = dev_env("RTO_TRACE_GEOMETRY", 0) != 0;      // developer aid (dev builds)
        if (trace) {
            static int last[4] = { -1, -1, -1, -1 };
            const int now[4] = { Q.boxX0, Q.boxY0, Q.boxW, Q.boxH };
            if (std::memcmp(last, now, sizeof now) != 0) {
                std::memcpy(last, now, sizeof now);
                std::fprintf(stderr, "[rto] launch geometry: rectangle x %d..%d y %d..%d, tile box %d,%d %dx%d = %d waves (+ fill: %d launched)\n",
                             rect[0], rect[2], rect[1], rect[3], Q.boxX0, Q.boxY0, Q.boxW, Q.boxH, Q.traceWaves, Q.launchWaves);
            }
        }
    }
    if (o) {
        rto_context::OrderState::Table& T = o->tab[o->active];
        const int box[4] = { Q.boxX0, Q.boxY0, Q.boxW, Q.boxH };
        // The table is a scheduling hint rebuilt from the costs the previous frames recorded: when the box changed its
        // SIZE (the entries are relative to the box's corner: a box that only moved -- a camera in motion -- keeps using
        // it) or every orderPeriod-th frame.  k_order_build emits a permutation of the box's tiles whatever the
        // cost array holds and keeps no state between calls: inside a capture it becomes a node of the graph.
        // A rebuild for age alone (same box size: a camera that stands still or only pans) doubles the next interval, up to 8
        // periods: the costs of such frames change slowly, and a build (13 us at config 2) every 8th frame was 4 % of a frame.
        const bool resized = !T.valid || box[2] != T.box[2] || box[3] != T.box[3];
        // A camera in motion (single-frame launches; the frames of a batch share one table): rebuilt every kMovedPeriod-th frame.  What
        // a table that is a few frames old gets wrong is not the rank of the tiles with work -- of config 2's 436 tiles of 40 trips
        // and more, 4 had fewer one frame (0.01 rad) earlier -- but the tiles the moving mask newly covers, which it holds among the
        // work-less ones at the end; the rim (tile_may_hit) takes care of those.  Orbit of 0.01 rad per frame, us per frame, table
        // rebuilt every 1st / 4th / 8th frame: 38.6 / 36.4 / 36.4 with the rim, 40.5 / 38.5 / 38.6 without; a build is 4.7 us + a
        // launch gap.  A camera that stands still backs off as before.
        bool moved = false;
        if (single && frameMode) {
            float cam[19];
            std::memcpy(cam, P.invView, 16 * sizeof(float)); std::memcpy(cam + 16, P.camPos, 3 * sizeof(float));
            moved = o->haveCam && std::memcmp(cam, o->lastCam, sizeof cam) != 0;
            std::memcpy(o->lastCam, cam, sizeof cam); o->haveCam = true;
        }
        static const int movedPeriod = (int)dev_env("RTO_ORDER_MOVED_PERIOD", kMovedPeriod);   // A/B knob (dev builds)
        if (moved && T.valid && T.age < movedPeriod) moved = false;
        if (!fixedOrder && Q.traceWaves > 0 && (resized || moved || T.age >= c->orderPeriod * T.stretch)) {
            T.stretch = (resized || moved) ? 1 : std::min(T.stretch * 2, 8);
            if (o->costValid) {
                const size_t stage = (size_t)(((Q.traceWaves + kOrderGroups - 1) / kOrderGroups + 15) & ~15);      // one byte per tile of a workgroup's sample
                if (stage > 60 * 1024) return fail(c, RTO_E_UNSUPPORTED, "render: frame too large for the launch-order sort (more than 983,040 tiles in the geometry's box)");
                static const int groups = []() { const int g = (int)dev_env("RTO_ORDER_GROUPS", kOrderGroups); return (g == 16 || g == 32 || g == 64) ? g : kOrderGroups; }();   // A/B knob (dev builds)
                hipLaunchKernelGGL(k_order_build, dim3(groups), dim3(kOrderBlock * kOrderGroups / groups), stage, s, o->d_tileCost, Q.tilesX,
                                   Q.boxX0, Q.boxY0, Q.boxW, Q.boxH, T.d, c->d_sortViolations);
                std::memcpy(T.box, box, sizeof box);
                T.valid = true; T.age = 0;
            } else T.valid = false;
        }
        Q.tileOrder = (T.valid && Q.traceWaves > 0) ? T.d : nullptr;
        if (frameMode && !fixedOrder) { Q.tileCost = o->d_tileCost; o->costValid = true; }
        if (frameMode) T.age++;
    }
    return RTO_OK;
}

// Threads per workgroup of the lean kernels (path 0: octree frames, 1: triangle frames; single and batch forms).  Their waves
// share nothing but the launch, so a workgroup is only a unit of dispatch: the dispatcher places a workgroup when ALL its waves
// fit, and with 4-wave workgroups a CU's wave slots stand empty until four are free at once (config 5's timeline: ~4,300 of
// 5,120 slots taken in mid-frame).  Measured (us per frame, 256 / 128 / 64 threads): config 5 453 / 445 / 439; config 2 36.5 /
// 43.2 / 43.3, config 4 75.1 / 81.2 / 78.5 -- the octree frames, whose launch geometry was tuned around 4-wave workgroups (four
// consecutive launch slots per CU, 4 -- since round 5: 5 -- resident waves per SIMD through the LDS request), keep 256.
// RTO_LEAN_BLOCK / RTO_TRI_BLOCK = <64|128|256> override (A/B runs).
static int lean_block(int path) {
    static const int b[2] = {
        []() { const int v = (int)dev_env("RTO_LEAN_BLOCK", 0); return (v == 64 || v == 128 || v == 256) ? v : 256; }(),
        []() { const int v = (int)dev_env("RTO_TRI_BLOCK", 0); return (v == 64 || v == 128 || v == 256) ? v : 64; }() };
    return b[path == 1 ? 1 : 0];
}
static int lean_wpb(int path) { return lean_block(path) / kWave; }

// Resident waves per SIMD of a lean octree launch, set through its dynamic LDS request (4-wave workgroups: workgroups per CU ==
// waves per SIMD).  With the occupancy mask a frame has fewer waves with work than the machine has wave slots at 6 per SIMD
// (config 2: ~5,800 against 6,144): all of them would be resident from the first microsecond, statically spread, and the
// frame would end when the unluckiest SIMD ends (round 3's timeline: SIMD totals of 91 +- 30 loop trips, last SIMD at 42 us, mean
// 29).  With fewer the last part of the waves is handed out as slots free up -- to whichever SIMD is done first.  The request
// for n: more than a (n + 1)-th of the CU's 160 KB, so that n + 1 workgroups cannot share it and n can (measured through the
// timeline's wave ids: 40 and 32 KB give 4 per SIMD -- five 32-KB requests do not fit --, 27 KB gives 5, 23 KB and less 6, the
// kernel's register limit).  Round 3 chose 4 (44.9 -> 38.9 us); round 5, the loop a third cheaper: 4 / 5 / 6 = config 2 28.2 /
// 27.6 / 29.0 us (orbit 29.8 / 28.9 / 29.9), config 4 59.0 / 56.6 / 55.5 (tools/wps_sweep.sh).  Launches of several frames have
// waves to spare and take no padding.  RTO_WAVES_PER_SIMD=<n> overrides (A/B runs); 0 = never pad.
static size_t lds_for_occupancy(size_t lds, int wavesDefault) {
    static const int forced = (int)dev_env("RTO_WAVES_PER_SIMD", -1);
    const int waves = forced >= 0 ? forced : wavesDefault;
    if (waves <= 0) return lds;
    const size_t groupsPerCU = (size_t)waves * 4 / (size_t)lean_wpb(0);     // 4 SIMDs per CU
    const size_t perGroup = (((size_t)(160 * 1024) / (groupsPerCU + 1)) + 1024) & ~(size_t)1023;
    return std::max(lds, std::min<size_t>(perGroup, 64 * 1024));
}

template <int MODE>
static int launch_trace(rto_context* c, const RenderParams& P, float4* d_out, hipStream_t s) {
    if (s != c->stream) c->otherStreams = true;
    { const int rcOrder = order_after_cull(c, s); if (rcOrder != RTO_OK) return rcOrder; }
    const int tiles = P.tilesX * P.tilesY;
    if (tiles <= 0) return RTO_OK;
    const int blocks = (tiles + (kBlock / kWave) - 1) / (kBlock / kWave);
    const bool packed = c->kernelMode >= RTO_KERNEL_PACKED || (c->kernelMode == RTO_KERNEL_AUTO && c->canonical);
    if (packed && !c->canonical) return fail(c, RTO_E_UNSUPPORTED, "render: packed kernel needs a canonical BFS octree");
    const bool capturing = stream_is_capturing(s);
    if (capturing && s != c->stream) c->foreignCaptured = true;
    const bool noEvents = capturing || c->eventsOff;     // events inside a capture cannot be timed; each record costs ~2 us between launches
    bool stopRecorded = false;
    hipEvent_t evA = c->ev0, evB = c->ev1;
    // a timing-ring slot is only taken by a launch that records events (events inside a capture cannot be timed)
    if (!noEvents && c->ringUsed < c->ringStart.size()) { evA = c->ringStart[c->ringUsed]; evB = c->ringStop[c->ringUsed]; c->ringUsed++; }
    bool startRecorded = false;
    // The lean kernels take the result of a frustum update from the device (StartState: visible at all? start at the root
    // or -- root culled, descendants visible, S/RT:765-812 -- at the first visible node): nothing to know on the host.
    // Every other kernel needs the host's copy; for them the culled-root edge goes to the generic kernel over the compacted array.
    const bool leanKernel = packed && c->canonical && (c->kernelMode == RTO_KERNEL_AUTO || c->kernelMode == RTO_KERNEL_PACKED || c->kernelMode == RTO_KERNEL_PACKED_PERSISTENT);
    const bool deviceStart = leanKernel && c->culling && c->cullAsync;
    if (c->culling && !deviceStart) { const int rcState = sync_cull_state(c); if (rcState != RTO_OK) return rcState; }
    const bool rootCulledEdge = !deviceStart && c->culling && !c->rootVisible && c->visibleNodes > 0;
    if (packed && !rootCulledEdge) {
        const size_t ldsStacks = (size_t)(kBlock / kWave) * (P.depth + 1) * kWave * sizeof(uint2);   // +1: the lean kernel's dummy entry
        if (c->kernelMode == RTO_KERNEL_PACKED_V1) {
            if (!noEvents) RTO_HIP(c, hipEventRecord(evA, s));
            startRecorded = true;
            RenderParams Q = P;
            Q.rootVisible = c->rootVisible;                 // as of sync_cull_state above
            hipLaunchKernelGGL(k_trace_packed<MODE>, dim3(blocks), dim3(kBlock), ldsStacks, s, Q, c->d_desc, d_out, c->d_steps, c->d_counters);
        } else {
            RenderParams Q = P;
            if (deviceStart) { Q.start = c->d_start; Q.rootVisible = 1; }       // what is visible is the device's knowledge: waves for the whole rectangle
            else Q.rootVisible = c->rootVisible;
            const bool frameMode = MODE == kModeColor || MODE == kModeShade;
            rto_context::OrderState* st = nullptr;
            const int solidRect[4] = { P.solidX0, P.solidY0, P.solidX1, P.solidY1 };
            {
                const bool maskable = leanKernel && c->kernelMode != RTO_KERNEL_PACKED_PERSISTENT;
                const int rc = prepare_schedule(c, s, capturing, frameMode, MODE == kModeTimeline, 0, solidRect, Q, &st, maskable ? 0 : -1, true);
                if (rc != RTO_OK) return rc;
            }
            const int lblocks = (Q.launchWaves + (kBlock / kWave) - 1) / (kBlock / kWave);
            const int lblocksLean = (Q.launchWaves + lean_wpb(0) - 1) / lean_wpb(0);
            const size_t ldsLean = (size_t)lean_wpb(0) * (P.depth + 1) * kWave * sizeof(uint2);
            // 5 resident waves per SIMD when the frame's waves with work would all be resident at once at 6 (see lds_for_occupancy): about
            // 40 % of the box's tiles have work behind the mask, so up to 3 x the machine's slots at 6 per SIMD.
            const bool fewWaves = Q.traceWaves <= 3 * 6 * 4 * c->numCUs;
            const size_t lds = Q.tileMask ? lds_for_occupancy(ldsLean, fewWaves ? 5 : 0) : (leanKernel && !(c->kernelMode == RTO_KERNEL_PACKED_PERSISTENT) ? lds_for_occupancy(ldsLean, 0) : ldsStacks);
            Q.maskLdsBytes = (int)lds;
            if (!noEvents) RTO_HIP(c, hipEventRecord(evA, s));        // after the order kernel: the pair brackets the traversal kernel alone
            startRecorded = true;
            const bool persistent = c->kernelMode == RTO_KERNEL_PACKED_PERSISTENT && st && frameMode;
            if (persistent) {
                if (!st->d_queue) {
                    if (capturing)
                        return fail(c, RTO_E_UNSUPPORTED, "render: the first persistent-kernel frame on a stream allocates its slot counter; "
                                                          "render one frame before hipStreamBeginCapture");
                    RTO_HIP(c, hipMalloc(&st->d_queue, sizeof(int)));
                }
                // every launch zeroes its own counter (a memset node when captured): replaying a graph of any number of
                // frames finds the same state each time
                RTO_HIP(c, hipMemsetAsync(st->d_queue, 0, sizeof(int), s));
                // enough workgroups to fill the machine at this kernel's occupancy; the rest of the slots come from the counter
                const int resident = c->numCUs * RTO_PERSIST_WAVES;
                hipLaunchKernelGGL(k_trace_lean_persistent<MODE>, dim3(std::min(lblocks, resident)), dim3(kBlock), lds, s, Q, c->d_desc, d_out,
                                   c->d_steps, c->d_counters, st->d_queue);
            } else if (c->kernelMode == RTO_KERNEL_PACKED_V3) {
                hipLaunchKernelGGL(k_trace_packed3<MODE>, dim3(lblocks), dim3(kBlock), lds, s, Q, c->d_desc, d_out, c->d_steps, c->d_counters);
            } else {
                // maskMode 2 (tests): a launch of the mask workgroups alone first, so that every wave of the frame finds the mask complete
                // two builds of the kernel, chosen by the scene's proof (P.exactGrid): each loop holds only its own form's child tests --
                // a variant more in that loop costs every wave (round 5: two more copies behind wave-uniform branches: +4 %)
                auto* kern = Q.exactGrid ? k_trace_lean<MODE, 1> : k_trace_lean<MODE, 0>;
                if (c->maskMode == 2 && Q.maskBlocks > 0)
                    hipLaunchKernelGGL(kern, dim3(Q.maskBlocks), dim3(lean_block(0)), lds, s, Q, c->d_desc, d_out, c->d_steps, c->d_counters);
                hipLaunchKernelGGL(kern, dim3(lblocksLean + Q.maskBlocks), dim3(lean_block(0)), lds, s, Q, c->d_desc, d_out, c->d_steps, c->d_counters);
            }
        }
    } else {
        const int rcCompact = ensure_compact(c, s);
        if (rcCompact != RTO_OK) return rcCompact;
        const rto_node* nodes = c->culling ? c->d_compact : c->d_nodes;
        RenderParams Q = P;
        if (c->culling && c->visibleNodes == 0) Q.rootVisible = 0;   // empty SSBO: nothing to traverse
        else if (c->culling) Q.rootVisible = 1;                      // compacted index 0 is whatever survived first (S/RT:765-772)
        if (!noEvents) RTO_HIP(c, hipEventRecord(evA, s));
        startRecorded = true;
        hipLaunchKernelGGL(k_trace_generic<MODE>, dim3(blocks), dim3(kBlock), 0, s, Q, nodes, d_out, c->d_steps, c->d_counters);
    }
    RTO_HIP(c, hipGetLastError());
    (void)startRecorded;
    if (!stopRecorded && !noEvents) RTO_HIP(c, hipEventRecord(evB, s));
    if (evA != c->ev0) { c->lastA = evA; c->lastB = evB; } else { c->lastA = c->ev0; c->lastB = c->ev1; }
    c->timed = !noEvents;
    return RTO_OK;
}

static int ensure_frame(rto_context* c, size_t pixels) {
    c->residentW = c->residentH = 0;          // whoever asks for the buffer is about to overwrite it
    if (c->frameCap >= pixels) return RTO_OK;
    (void)hipFree(c->d_frame); c->d_frame = nullptr; c->frameCap = 0;
    RTO_HIP(c, hipMalloc(&c->d_frame, pixels * sizeof(float4)));
    c->frameCap = pixels;
    return RTO_OK;
}

static int ensure_steps(rto_context* c, size_t pixels) {
    if (c->stepsCap >= pixels) return RTO_OK;
    (void)hipFree(c->d_steps); c->d_steps = nullptr; c->stepsCap = 0;
    RTO_HIP(c, hipMalloc(&c->d_steps, pixels * sizeof(int)));
    c->stepsCap = pixels;
    return RTO_OK;
}

// Records of the lean triangle kernel (k_unified_*): after k_desc_trimask, on stream s.  Leaves c->d_triRec.
static int build_triangle_records(rto_context* c, hipStream_t s) {
    (void)hipFree(c->d_triRec); c->d_triRec = nullptr;
    if (!(c->canonical && c->numInternal > 0)) return RTO_OK;
    const int64_t n = c->numInternal;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    BuildScratch scratch(s);
    int *d_count = nullptr, *d_first = nullptr, *d_bs = nullptr, *d_bb = nullptr;
    int64_t* d_total = nullptr;
    RTO_HIP(c, scratch.alloc(&d_count, (size_t)n)); RTO_HIP(c, scratch.alloc(&d_first, (size_t)n + 1));
    RTO_HIP(c, scratch.alloc(&d_bs, (size_t)nb)); RTO_HIP(c, scratch.alloc(&d_bb, (size_t)nb)); RTO_HIP(c, scratch.alloc(&d_total, 1));
    hipLaunchKernelGGL(k_unified_count, dim3(nb), dim3(kBlock), 0, s, c->d_desc, n, d_count);
    hipLaunchKernelGGL(k_block_sums, dim3(nb), dim3(kBlock), 0, s, d_count, n, d_bs);
    hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, s, d_bs, nb, d_bb, d_total);
    hipLaunchKernelGGL(k_block_exclusive_scan, dim3(nb), dim3(kBlock), 0, s, d_count, n, d_bb, d_total, d_first);
    RTO_HIP(c, hipGetLastError());
    int64_t total = 0;
    RTO_HIP(c, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    if (total + 1 > 0x7fffffff) return fail(c, RTO_E_UNSUPPORTED, "leaf triangles: record indices are int32");
    RTO_HIP(c, hipMalloc(&c->d_triRec, (size_t)(total + 1) * sizeof(uint2)));
    hipLaunchKernelGGL(k_unified_fill, dim3(nb), dim3(kBlock), 0, s, c->d_desc, c->d_descFirstChild, c->d_triOffset, d_first, n, c->d_triRec);
    RTO_HIP(c, hipGetLastError());
    return RTO_OK;
}

// The batch kernels put (mask workgroups per frame) x (frames) workgroups in front of the grid: every frame of a batch brings
// its mask, or none does (a frame skipped by the loop above, a degenerate projection).
static void batch_mask_all_or_none(RenderBatch& B) {
    bool all = true;
    for (int i = 0; i < B.n; i++) all = all && B.P[i].tileMask != nullptr && B.P[i].maskBlocks == B.P[0].maskBlocks;
    if (all) return;
    for (int i = 0; i < B.n; i++) { B.P[i].tileMask = nullptr; B.P[i].maskBlocks = 0; }
}

// n frames (n <= kMaxBatch) in one launch of k_trace_lean_batch; falls back to n launches when the lean kernel is not the
// one in use (generic array, another kernel selected, the culled-root edge).  MODE: kModeColor or kModeShade.
template <int MODE>
static int launch_trace_batch(rto_context* c, const RenderParams* Ps, int n, float4* const* outs, hipStream_t s) {
    if (s != c->stream) c->otherStreams = true;
    const bool packed = c->kernelMode >= RTO_KERNEL_PACKED || (c->kernelMode == RTO_KERNEL_AUTO && c->canonical);
    const bool lean = packed && c->canonical && (c->kernelMode == RTO_KERNEL_AUTO || c->kernelMode == RTO_KERNEL_PACKED) &&
                      (!c->culling || c->cullAsync);
    bool sameDepth = true;
    for (int i = 1; i < n; i++) sameDepth = sameDepth && Ps[i].depth == Ps[0].depth;
    if (!lean || n == 1 || !sameDepth) {
        for (int i = 0; i < n; i++) {
            const int rc = launch_trace<MODE>(c, Ps[i], outs[i], s);
            if (rc != RTO_OK) return rc;
        }
        return RTO_OK;
    }
    const bool capturing = stream_is_capturing(s);
    if (capturing && s != c->stream) c->foreignCaptured = true;
    { const int rcOrder = order_after_cull(c, s); if (rcOrder != RTO_OK) return rcOrder; }
    const bool noEvents = capturing || c->eventsOff;
    RenderBatch B;
    B.n = n;
    int maxWaves = 0;
    rto_context::OrderState* st = nullptr;
    for (int i = 0; i < n; i++) {
        B.P[i] = Ps[i];
        if (c->culling) { B.P[i].start = c->d_start; B.P[i].rootVisible = 1; }     // the frustum update's result stays on the device (launch_trace)
        B.out[i] = outs[i];
        if (Ps[i].tilesX * Ps[i].tilesY <= 0) { B.P[i].launchWaves = 0; continue; }
        const int solidRect[4] = { Ps[i].solidX0, Ps[i].solidY0, Ps[i].solidX1, Ps[i].solidY1 };
        const int rc = prepare_schedule(c, s, capturing, true, false, 0, solidRect, B.P[i], &st, i);
        if (rc != RTO_OK) return rc;
        maxWaves = std::max(maxWaves, B.P[i].launchWaves);
    }
    batch_mask_all_or_none(B);
    // the frames share this stream's launch-order table: its entries are relative to the box's corner, so it serves every
    // frame whose box has the size of the one it was last rebuilt for (a moving camera shifts the box far more often than
    // it resizes it); any other box falls back to the centre-out order (any permutation of a frame's own box renders it)
    for (int i = 0; i < n; i++) {
        if (!st || B.P[i].launchWaves <= 0 || !B.P[i].tileCost) continue;       // tileCost set: the temporal order is in use for this frame
        const rto_context::OrderState::Table& T = st->tab[st->active];
        const bool fits = T.valid && B.P[i].boxW == T.box[2] && B.P[i].boxH == T.box[3] && B.P[i].traceWaves > 0;
        B.P[i].tileOrder = fits ? T.d : nullptr;
    }
    if (maxWaves <= 0) return RTO_OK;
    hipEvent_t evA = c->ev0, evB = c->ev1;
    if (!noEvents && c->ringUsed < c->ringStart.size()) { evA = c->ringStart[c->ringUsed]; evB = c->ringStop[c->ringUsed]; c->ringUsed++; }
    if (!noEvents) RTO_HIP(c, hipEventRecord(evA, s));
    const size_t lds = lds_for_occupancy((size_t)lean_wpb(0) * (Ps[0].depth + 1) * kWave * sizeof(uint2), 0);
    for (int i = 0; i < n; i++) B.P[i].maskLdsBytes = (int)lds;
    const long long waves = (long long)maxWaves * n;
    // the batch kernel keeps the run-time flag: with waves of several frames to fill every SIMD, its one build (all five child-test forms
    // in the loop) measured 1.5-3 % FASTER than the two builds that pay for single frames (4 frames per launch: 0.0259 against 0.0270 ms)
    auto* kern = k_trace_lean_batch<MODE, -1>;
    if (c->maskMode == 2 && B.P[0].maskBlocks > 0)
        hipLaunchKernelGGL(kern, dim3((unsigned)(B.P[0].maskBlocks * n)), dim3(lean_block(0)), lds, s, B, c->d_desc);
    hipLaunchKernelGGL(kern, dim3((unsigned)((waves + lean_wpb(0) - 1) / lean_wpb(0)) + (unsigned)(B.P[0].maskBlocks * n)), dim3(lean_block(0)), lds, s, B, c->d_desc);
    RTO_HIP(c, hipGetLastError());
    if (!noEvents) RTO_HIP(c, hipEventRecord(evB, s));
    c->lastA = evA; c->lastB = evB;                       // what rto_last_kernel_ms reads
    c->timed = !noEvents;
    return RTO_OK;
}

extern "C" {

int rto_render_device(rto_context* c, const rto_frame* f, const rto_partition* p, void* d_out, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!d_out) return fail(c, RTO_E_INVALID, "rto_render_device: d_out is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    RenderParams P;
    int rc = fill_params(c, f, p, P, (hipStream_t)hip_stream);
    if (rc != RTO_OK) return rc;
    return launch_trace<kModeColor>(c, P, (float4*)d_out, (hipStream_t)hip_stream);
}

int rto_render_shade_device(rto_context* c, const rto_frame* f, const rto_partition* p, void* d_shade, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!d_shade) return fail(c, RTO_E_INVALID, "rto_render_shade_device: d_shade is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    RenderParams P;
    int rc = fill_params(c, f, p, P, (hipStream_t)hip_stream);
    if (rc != RTO_OK) return rc;
    return launch_trace<kModeShade>(c, P, (float4*)d_shade, (hipStream_t)hip_stream);
}

int rto_render_host(rto_context* c, const rto_frame* f, float* host_rgba) {
    if (!c) return RTO_E_INVALID;
    if (!host_rgba) return fail(c, RTO_E_INVALID, "rto_render_host: host_rgba is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    RenderParams P;
    int rc = fill_params(c, f, nullptr, P);
    if (rc != RTO_OK) return rc;
    const size_t pixels = (size_t)f->width * f->height;
    if ((rc = ensure_frame(c, pixels)) != RTO_OK) return rc;
    if ((rc = launch_trace<kModeColor>(c, P, c->d_frame, c->stream)) != RTO_OK) return rc;
    RTO_HIP(c, hipMemcpyAsync(host_rgba, c->d_frame, pixels * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    return RTO_OK;
}

static int assemble_common(rto_context* c, const rto_frame* f, const rto_partition* p, const void* d_gathered, int batch, int index,
                           void* d_frame, void* hip_stream, bool shade, const char* who) {
    if (!c || !f || !p || !d_gathered || !d_frame) return c ? fail(c, RTO_E_INVALID, std::string(who) + ": NULL argument") : RTO_E_INVALID;
    if (p->num_parts < 1 || p->band_rows <= 0) return fail(c, RTO_E_INVALID, std::string(who) + ": bad partition");
    if (batch < 1 || index < 0 || index >= batch) return fail(c, RTO_E_INVALID, std::string(who) + ": index must lie in [0, batch)");
    RTO_HIP(c, hipSetDevice(c->device));
    rto_partition p0 = *p; p0.part = 0;
    const size_t partPixels = (size_t)rto_partition_rows(f, &p0) * (size_t)f->width;
    const size_t stride = partPixels * (size_t)batch, base = partPixels * (size_t)index;
    const int bandRows = p->num_parts > 1 ? p->band_rows : f->height;
    if (shade)
        hipLaunchKernelGGL(k_assemble_shade, dim3(1, (unsigned)f->height, 1), dim3(256), 0, (hipStream_t)hip_stream,
                           (const float*)d_gathered + base, (char*)d_frame, (size_t)0, (size_t)0, f->width, f->height, p->num_parts, bandRows, stride);
    else
        hipLaunchKernelGGL(k_assemble, dim3(1, (unsigned)f->height, 1), dim3(256), 0, (hipStream_t)hip_stream,
                           (const float4*)d_gathered + base, (char*)d_frame, (size_t)0, (size_t)0, f->width, f->height, p->num_parts, bandRows, stride);
    RTO_HIP(c, hipGetLastError());
    return RTO_OK;
}

int rto_assemble_device(rto_context* c, const rto_frame* f, const rto_partition* p, const void* d_gathered, void* d_frame, void* hip_stream) {
    return assemble_common(c, f, p, d_gathered, 1, 0, d_frame, hip_stream, false, "rto_assemble_device");
}

int rto_assemble_shade_device(rto_context* c, const rto_frame* f, const rto_partition* p, const void* d_gathered, void* d_frame, void* hip_stream) {
    return assemble_common(c, f, p, d_gathered, 1, 0, d_frame, hip_stream, true, "rto_assemble_shade_device");
}

int rto_assemble_batch_device(rto_context* c, const rto_frame* f, const rto_partition* p, const void* d_gathered, int batch, int index,
                              int shade_payload, void* d_frame, void* hip_stream) {
    return assemble_common(c, f, p, d_gathered, batch, index, d_frame, hip_stream, shade_payload != 0, "rto_assemble_batch_device");
}

int rto_render_batch_device(rto_context* c, const rto_frame* frames, int n, const rto_partition* p, int shade_payload, void* d_out,
                            size_t frame_stride_bytes, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!frames || n < 1 || !d_out) return fail(c, RTO_E_INVALID, "rto_render_batch_device: NULL argument / empty batch");
    RTO_HIP(c, hipSetDevice(c->device));
    for (int i = 0; i < n; i++)
        if (frames[i].width != frames[0].width || frames[i].height != frames[0].height)
            return fail(c, RTO_E_INVALID, "rto_render_batch_device: the frames of a batch share width and height");
    for (int i0 = 0; i0 < n; i0 += kMaxBatch) {
        const int m = std::min(kMaxBatch, n - i0);
        RenderParams P[kMaxBatch];
        float4* outs[kMaxBatch];
        for (int i = 0; i < m; i++) {
            const int rc = fill_params(c, &frames[i0 + i], p, P[i], (hipStream_t)hip_stream);
            if (rc != RTO_OK) return rc;
            outs[i] = reinterpret_cast<float4*>(static_cast<char*>(d_out) + (size_t)(i0 + i) * frame_stride_bytes);
        }
        const int rc = shade_payload ? launch_trace_batch<kModeShade>(c, P, m, outs, (hipStream_t)hip_stream)
                                     : launch_trace_batch<kModeColor>(c, P, m, outs, (hipStream_t)hip_stream);
        if (rc != RTO_OK) return rc;
    }
    return RTO_OK;
}

int rto_assemble_batch_all_device(rto_context* c, const rto_frame* frames, int batch, const rto_partition* p, const void* d_gathered,
                                  int shade_payload, void* d_frames, size_t frame_stride_bytes, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!frames || batch < 1 || !d_frames) return fail(c, RTO_E_INVALID, "rto_assemble_batch_all_device: NULL argument / empty batch");
    for (int i = 0; i < batch; i++) {
        const int rc = assemble_common(c, &frames[i], p, d_gathered, batch, i, static_cast<char*>(d_frames) + (size_t)i * frame_stride_bytes,
                                       hip_stream, shade_payload != 0, "rto_assemble_batch_all_device");
        if (rc != RTO_OK) return rc;
    }
    return RTO_OK;
}

static int run_steps(rto_context* c, const rto_frame* f, rto_stats* st, int32_t* host_steps) {
    RTO_HIP(c, hipSetDevice(c->device));
    RenderParams P;
    int rc = fill_params(c, f, nullptr, P);
    if (rc != RTO_OK) return rc;
    const size_t pixels = (size_t)f->width * f->height;
    if ((rc = ensure_steps(c, pixels)) != RTO_OK) return rc;
    RTO_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    RTO_HIP(c, hipMemsetAsync(c->d_steps, 0, pixels * sizeof(int), c->stream));
    if ((rc = launch_trace<kModeSteps>(c, P, nullptr, c->stream)) != RTO_OK) return rc;
    Counters h;
    RTO_HIP(c, hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    if (host_steps) RTO_HIP(c, hipMemcpyAsync(host_steps, c->d_steps, pixels * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    if (st) { st->rays = pixels; st->pops = h.pops; st->hits = h.hits; st->capped = h.capped; }
    return RTO_OK;
}

int rto_frame_stats(rto_context* c, const rto_frame* f, rto_stats* out) {
    if (!c) return RTO_E_INVALID;
    if (!out) return fail(c, RTO_E_INVALID, "rto_frame_stats: out is NULL");
    return run_steps(c, f, out, nullptr);
}

int rto_render_steps_host(rto_context* c, const rto_frame* f, int32_t* host_steps) {
    if (!c) return RTO_E_INVALID;
    if (!host_steps) return fail(c, RTO_E_INVALID, "rto_render_steps_host: host_steps is NULL");
    return run_steps(c, f, nullptr, host_steps);
}

int rto_debug_timeline(rto_context* c, const rto_frame* f, int32_t* host_records, int64_t capacity_tiles, int64_t* num_tiles) {
    if (!c || !f || !num_tiles) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    if (!c->canonical) return fail(c, RTO_E_UNSUPPORTED, "rto_debug_timeline: packed kernel only");
    RenderParams P;
    int rc = fill_params(c, f, nullptr, P);
    if (rc != RTO_OK) return rc;
    const int64_t tiles = (int64_t)P.tilesX * P.tilesY;
    *num_tiles = tiles;
    if (!host_records) return RTO_OK;
    if (capacity_tiles < tiles) return fail(c, RTO_E_INVALID, "rto_debug_timeline: capacity too small");
    const size_t pixels = (size_t)f->width * f->height;
    if ((rc = ensure_frame(c, pixels)) != RTO_OK) return rc;
    if ((rc = ensure_steps(c, (size_t)tiles * 8 > pixels ? (size_t)tiles * 8 : pixels)) != RTO_OK) return rc;
    const int saved = c->kernelMode;
    if (saved < RTO_KERNEL_PACKED) c->kernelMode = RTO_KERNEL_PACKED;
    RTO_HIP(c, hipMemsetAsync(c->d_steps, 0, (size_t)tiles * 8 * sizeof(int), c->stream));   // tiles without a wave: all-zero records
    rc = launch_trace<kModeTimeline>(c, P, c->d_frame, c->stream);
    c->kernelMode = saved;
    if (rc != RTO_OK) return rc;
    RTO_HIP(c, hipMemcpyAsync(host_records, c->d_steps, (size_t)tiles * 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    return RTO_OK;
}

int rto_upload_leaf_triangles(rto_context* c, const float* tris, int64_t num_tris, const int32_t* tri_offset) {
    if (!c) return RTO_E_INVALID;
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_upload_leaf_triangles: upload the octree first");
    if (!tri_offset || num_tris < 0 || (num_tris > 0 && !tris)) return fail(c, RTO_E_INVALID, "rto_upload_leaf_triangles: NULL argument");
    if (tri_offset[0] != 0 || tri_offset[c->numNodes] != num_tris)
        return fail(c, RTO_E_INVALID, "rto_upload_leaf_triangles: tri_offset must hold numNodes+1 entries running from 0 to num_tris");
    for (int64_t i = 0; i < c->numNodes; i++)
        if (tri_offset[i + 1] < tri_offset[i]) return fail(c, RTO_E_INVALID, "rto_upload_leaf_triangles: tri_offset must be non-decreasing");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());          // frames in flight on any stream still read the old buffers / descriptors
    (void)hipFree(c->d_tris); c->d_tris = nullptr;
    (void)hipFree(c->d_triOffset); c->d_triOffset = nullptr;
    (void)hipFree(c->d_triRec); c->d_triRec = nullptr;
    RTO_HIP(c, hipMalloc(&c->d_tris, (size_t)(num_tris ? num_tris : 1) * 12 * sizeof(float)));
    RTO_HIP(c, hipMalloc(&c->d_triOffset, (size_t)(c->numNodes + 1) * sizeof(int)));
    if (num_tris) RTO_HIP(c, hipMemcpy(c->d_tris, tris, (size_t)num_tris * 12 * sizeof(float), hipMemcpyHostToDevice));
    RTO_HIP(c, hipMemcpy(c->d_triOffset, tri_offset, (size_t)(c->numNodes + 1) * sizeof(int), hipMemcpyHostToDevice));
    c->numTris = num_tris;
    if (c->canonical && c->numInternal > 0) {     // third child mask of the packed form: leaf children that own triangles
        hipLaunchKernelGGL(k_desc_trimask, dim3((unsigned)((c->numInternal + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           c->d_descFirstChild, c->d_triOffset, c->numInternal, c->d_desc);
        RTO_HIP(c, hipGetLastError());
        const int rcRec = build_triangle_records(c, c->stream);
        if (rcRec != RTO_OK) return rcRec;
        RTO_HIP(c, hipStreamSynchronize(c->stream));
    }
    return RTO_OK;
}

}  // extern "C"

namespace {
#include "../host/mc_cases.inc"
// 256 cases -> triangle count << 60 | edge nibbles (first edge lowest)
void pack_mc_cases(unsigned long long out[256]) {
    for (int i = 0; i < 256; i++) {
        unsigned long long v = 0;
        int n = 0;
        for (const char* e = kMcCaseEdges[i]; *e != 'f'; e++, n++) {
            const unsigned long long d = (unsigned long long)(*e <= '9' ? *e - '0' : *e - 'a' + 10);
            v |= d << (4 * n);
        }
        out[i] = v | ((unsigned long long)(n / 3) << 60);
    }
}
}  // namespace

extern "C" {

int rto_build_leaf_triangles(rto_context* c, const uint8_t* voxels, int dimX, int dimY, int dimZ) {
    if (!c) return RTO_E_INVALID;
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_build_leaf_triangles: upload or build the octree first");
    RTO_HIP(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    RTO_HIP(c, hipDeviceSynchronize());          // frames in flight on any stream still read the old buffers / descriptors
    BuildScratch scratch(c->stream);
    hipEvent_t e0, e1, e2;
    RTO_HIP(c, hipEventCreate(&e0)); RTO_HIP(c, hipEventCreate(&e1)); RTO_HIP(c, hipEventCreate(&e2));
    struct EvGuard { hipEvent_t a, b, d; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(d); } } evg{ e0, e1, e2 };
    const uint8_t* d_vox = nullptr;
    RTO_HIP(c, hipEventRecord(e0, s));
    if (voxels) {
        if (dimX <= 0 || dimY <= 0 || dimZ <= 0) return fail(c, RTO_E_INVALID, "rto_build_leaf_triangles: bad grid dimensions");
        int maxDim = dimX > dimY ? dimX : dimY;
        if (dimZ > maxDim) maxDim = dimZ;
        if (maxDim > c->rootSize) return fail(c, RTO_E_INVALID, "rto_build_leaf_triangles: the grid is larger than the resident octree's root");
        uint8_t* up = nullptr;
        RTO_HIP(c, scratch.alloc(&up, (size_t)dimX * dimY * dimZ));
        RTO_HIP(c, hipMemcpyAsync(up, voxels, (size_t)dimX * dimY * dimZ, hipMemcpyHostToDevice, s));
        d_vox = up;
    } else {
        if (!c->d_vox) return fail(c, RTO_E_INVALID, "rto_build_leaf_triangles: voxels == NULL needs an octree made by rto_build_octree");
        d_vox = c->d_vox; dimX = c->voxDim[0]; dimY = c->voxDim[1]; dimZ = c->voxDim[2];
    }
    RTO_HIP(c, hipEventRecord(e1, s));
    if (!c->d_mcCases) {
        unsigned long long packed[256];
        pack_mc_cases(packed);
        RTO_HIP(c, hipMalloc(&c->d_mcCases, sizeof packed));
        RTO_HIP(c, hipMemcpy(c->d_mcCases, packed, sizeof packed, hipMemcpyHostToDevice));
    }
    (void)hipFree(c->d_tris); c->d_tris = nullptr;
    (void)hipFree(c->d_triOffset); c->d_triOffset = nullptr;
    (void)hipFree(c->d_triRec); c->d_triRec = nullptr;
    c->numTris = 0;

    const int64_t n = c->numNodes;
    const int nb = (int)((n + kBlock - 1) / kBlock);
    LeafTriParams P{ c->d_nodes, n, d_vox, dimX, dimY, dimZ, c->gridMin[0], c->gridMin[1], c->gridMin[2], c->voxelSize, c->d_mcCases };
    int *d_count = nullptr, *d_big = nullptr, *d_bigCount = nullptr, *d_bs = nullptr, *d_bb = nullptr;
    int64_t* d_total = nullptr;
    RTO_HIP(c, scratch.alloc(&d_count, (size_t)n)); RTO_HIP(c, scratch.alloc(&d_big, (size_t)n)); RTO_HIP(c, scratch.alloc(&d_bigCount, 3));
    RTO_HIP(c, scratch.alloc(&d_bs, (size_t)nb)); RTO_HIP(c, scratch.alloc(&d_bb, (size_t)nb)); RTO_HIP(c, scratch.alloc(&d_total, 1));
    RTO_HIP(c, hipMalloc(&c->d_triOffset, (size_t)(n + 1) * sizeof(int)));
    RTO_HIP(c, hipMemsetAsync(d_bigCount, 0, 3 * sizeof(int), s));        // [0] big leaves, [1] their chunks, [2] chunk cursor
    // pass 1: counts
    hipLaunchKernelGGL(k_leaftri_small<false>, dim3(nb), dim3(kBlock), 0, s, P, d_count, (const int*)nullptr, d_big, d_bigCount, (float*)nullptr);
    RTO_HIP(c, hipGetLastError());
    int big[2] = { 0, 0 };
    RTO_HIP(c, hipMemcpyAsync(big, d_bigCount, sizeof big, hipMemcpyDeviceToHost, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    const int bigCount = big[0], numChunks = big[1];
    const int nbBig = (bigCount + kBlock - 1) / kBlock;
    const int nbChunk = (numChunks + (kBlock / kWave) - 1) / (kBlock / kWave);
    int *d_bigFirst = nullptr, *d_chunkCount = nullptr, *d_chunkOff = nullptr;
    int2* d_chunks = nullptr;
    if (bigCount > 0) {
        RTO_HIP(c, scratch.alloc(&d_bigFirst, (size_t)bigCount)); RTO_HIP(c, scratch.alloc(&d_chunks, (size_t)numChunks));
        RTO_HIP(c, scratch.alloc(&d_chunkCount, (size_t)numChunks)); RTO_HIP(c, scratch.alloc(&d_chunkOff, (size_t)numChunks));
        hipLaunchKernelGGL(k_leaftri_plan, dim3(nbBig), dim3(kBlock), 0, s, P, d_big, bigCount, d_bigCount + 2, d_bigFirst, d_chunks);
        hipLaunchKernelGGL(k_leaftri_big<false>, dim3(nbChunk), dim3(kBlock), 0, s, P, d_big, d_chunks, numChunks, d_chunkCount, (const int*)nullptr,
                           (const int*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(k_leaftri_bigsum, dim3(nbBig), dim3(kBlock), 0, s, P, d_big, bigCount, d_bigFirst, d_chunkCount, d_chunkOff, d_count);
    }
    // scan -> triOffset
    hipLaunchKernelGGL(k_block_sums, dim3(nb), dim3(kBlock), 0, s, d_count, n, d_bs);
    hipLaunchKernelGGL(k_scan_block_counts, dim3(1), dim3(1024), 0, s, d_bs, nb, d_bb, d_total);
    hipLaunchKernelGGL(k_block_exclusive_scan, dim3(nb), dim3(kBlock), 0, s, d_count, n, d_bb, d_total, c->d_triOffset);
    RTO_HIP(c, hipGetLastError());
    int64_t total = 0;
    RTO_HIP(c, hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    if (total > 0x7fffffff / 12) return fail(c, RTO_E_UNSUPPORTED, "rto_build_leaf_triangles: triangle offsets are int32");
    RTO_HIP(c, hipMalloc(&c->d_tris, (size_t)(total ? total : 1) * 12 * sizeof(float)));
    // pass 2: triangles
    hipLaunchKernelGGL(k_leaftri_small<true>, dim3(nb), dim3(kBlock), 0, s, P, d_count, c->d_triOffset, d_big, d_bigCount, c->d_tris);
    if (bigCount > 0) hipLaunchKernelGGL(k_leaftri_big<true>, dim3(nbChunk), dim3(kBlock), 0, s, P, d_big, d_chunks, numChunks, d_chunkCount, d_chunkOff, c->d_triOffset, c->d_tris);
    RTO_HIP(c, hipGetLastError());
    c->numTris = total;
    if (c->canonical && c->numInternal > 0) {
        hipLaunchKernelGGL(k_desc_trimask, dim3((unsigned)((c->numInternal + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           c->d_descFirstChild, c->d_triOffset, c->numInternal, c->d_desc);
        RTO_HIP(c, hipGetLastError());
        const int rcRec = build_triangle_records(c, s);
        if (rcRec != RTO_OK) return rcRec;
    }
    RTO_HIP(c, hipEventRecord(e2, s));
    RTO_HIP(c, hipStreamSynchronize(s));
    RTO_HIP(c, hipEventElapsedTime(&c->buildUploadMs, e0, e1));
    RTO_HIP(c, hipEventElapsedTime(&c->buildMs, e1, e2));
    return RTO_OK;
}

int rto_download_leaf_triangles(rto_context* c, float* tris, int64_t tri_capacity, int32_t* tri_offset, int64_t* num_tris) {
    if (!c || !num_tris) return RTO_E_INVALID;
    if (!c->d_triOffset) return fail(c, RTO_E_NO_OCTREE, "rto_download_leaf_triangles: no leaf triangles resident");
    *num_tris = c->numTris;
    if (!tris && !tri_offset) return RTO_OK;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    if (tris) {
        if (tri_capacity < c->numTris) return fail(c, RTO_E_INVALID, "rto_download_leaf_triangles: capacity too small");
        if (c->numTris) RTO_HIP(c, hipMemcpy(tris, c->d_tris, (size_t)c->numTris * 12 * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (tri_offset) RTO_HIP(c, hipMemcpy(tri_offset, c->d_triOffset, (size_t)(c->numNodes + 1) * sizeof(int), hipMemcpyDeviceToHost));
    return RTO_OK;
}

static int launch_triangles(rto_context* c, const rto_frame* f, const rto_partition* p, int shadow, float4* d_out, hipStream_t s,
                            bool count, bool shadeOut = false) {
    if (s != c->stream) c->otherStreams = true;
    if (!c->d_triOffset) return fail(c, RTO_E_NO_OCTREE, "render_triangles: no leaf triangles uploaded");
    if (c->culling) return fail(c, RTO_E_UNSUPPORTED, "render_triangles: not available while frustum culling is active");
    { const int rcOrder = order_after_cull(c, s); if (rcOrder != RTO_OK) return rcOrder; }      // a culling switch-off rewrites the descriptors too
    RenderParams P;
    int rc = fill_params(c, f, p, P, s);
    if (rc != RTO_OK) return rc;
    const int tiles = P.tilesX * P.tilesY;
    if (tiles <= 0) return RTO_OK;
    const int blocks = (tiles + (kBlock / kWave) - 1) / (kBlock / kWave);
    const bool packed = c->canonical && c->numInternal > 0 && c->kernelMode != RTO_KERNEL_GENERIC;
    const bool capturing = stream_is_capturing(s);
    if (packed) {
        // same launch geometry and launch order as the octree frames: waves for the geometry's rectangle only, wide stores
        // for the rest, costliest tiles of earlier frames first (the instrumented frame keeps one wave per tile)
        rto_context::OrderState* st = nullptr;
        const int solidRect[4] = { P.solidX0, P.solidY0, P.solidX1, P.solidY1 };
        const bool leanTri = c->d_triRec && c->kernelMode != RTO_KERNEL_PACKED_V3;
        if ((rc = prepare_schedule(c, s, capturing, !count, false, 1, solidRect, P, &st, leanTri ? 0 : -1, true)) != RTO_OK) return rc;
    }
    const bool noEvents = capturing || c->eventsOff;
    hipEvent_t evA = c->ev0, evB = c->ev1;                 // a timing-ring slot per launch that records events (rto_timing_begin)
    if (!noEvents && c->ringUsed < c->ringStart.size()) { evA = c->ringStart[c->ringUsed]; evB = c->ringStop[c->ringUsed]; c->ringUsed++; }
    if (!noEvents) RTO_HIP(c, hipEventRecord(evA, s));
    if (packed) {
        const int lblocks = (P.launchWaves + (kBlock / kWave) - 1) / (kBlock / kWave);
        if (c->d_triRec && c->kernelMode != RTO_KERNEL_PACKED_V3) {
            // default: the lean loop on the unified records (8-byte stack entries, shadow rays start inside the loop)
            LeanTriScene S{ c->d_triRec, c->d_tris };
#if defined(RTO_TRI_TIMELINE)
            S.timeline = (!count && !shadeOut && !capturing && ensure_steps(c, (size_t)tiles * 8) == RTO_OK) ? c->d_steps : nullptr;
#endif
#if defined(RTO_TRI_STAMP)
            S.stamp = (!count && !shadeOut && !capturing && ensure_steps(c, (size_t)tiles * 24) == RTO_OK) ? reinterpret_cast<unsigned*>(c->d_steps) : nullptr;
#endif
            const size_t lds = (size_t)lean_wpb(1) * ((P.depth + 1) * kWave * sizeof(uint2) + kWave * sizeof(unsigned long long));   // stacks + the keys of the triangle rounds
            P.maskLdsBytes = (int)lds;
            const dim3 lb(lean_block(1)), lgrid((P.launchWaves + lean_wpb(1) - 1) / lean_wpb(1) + P.maskBlocks);
            if (c->maskMode == 2 && P.maskBlocks > 0 && !count)
                hipLaunchKernelGGL((k_trace_lean_triangles<kModeColor, false>), dim3(P.maskBlocks), lb, lds, s, P, S, shadow, d_out, c->d_counters);
            if (shadeOut) hipLaunchKernelGGL((k_trace_lean_triangles<kModeColor, true>), lgrid, lb, lds, s, P, S, shadow, d_out, c->d_counters);
            else if (count) hipLaunchKernelGGL((k_trace_lean_triangles<kModeSteps, false>), lgrid, lb, lds, s, P, S, shadow, d_out, c->d_counters);
            else hipLaunchKernelGGL((k_trace_lean_triangles<kModeColor, false>), lgrid, lb, lds, s, P, S, shadow, d_out, c->d_counters);
        } else {
            // RTO_KERNEL_PACKED_V3: round 1's form on the descriptors + triOffset (kept in the test matrix)
            PackedTriScene S{ c->d_desc, c->d_descFirstChild, c->d_tris, c->d_triOffset };
            const size_t lds = (size_t)(kBlock / kWave) * P.depth * kWave * sizeof(uint4);
            if (shadeOut) hipLaunchKernelGGL((k_trace_packed_triangles<kModeColor, true>), dim3(lblocks), dim3(kBlock), lds, s, P, S, shadow, d_out, c->d_counters);
            else if (count) hipLaunchKernelGGL((k_trace_packed_triangles<kModeSteps, false>), dim3(lblocks), dim3(kBlock), lds, s, P, S, shadow, d_out, c->d_counters);
            else hipLaunchKernelGGL((k_trace_packed_triangles<kModeColor, false>), dim3(lblocks), dim3(kBlock), lds, s, P, S, shadow, d_out, c->d_counters);
        }
    } else {
        TriScene S{ c->d_nodes, c->d_tris, c->d_triOffset };
        if (shadeOut) hipLaunchKernelGGL((k_trace_triangles<kModeColor, true>), dim3(blocks), dim3(kBlock), 0, s, P, S, shadow, d_out, c->d_counters);
        else if (count) hipLaunchKernelGGL((k_trace_triangles<kModeSteps, false>), dim3(blocks), dim3(kBlock), 0, s, P, S, shadow, d_out, c->d_counters);
        else hipLaunchKernelGGL((k_trace_triangles<kModeColor, false>), dim3(blocks), dim3(kBlock), 0, s, P, S, shadow, d_out, c->d_counters);
    }
    RTO_HIP(c, hipGetLastError());
    if (!noEvents) RTO_HIP(c, hipEventRecord(evB, s));
    c->lastA = evA; c->lastB = evB;
    c->timed = !noEvents;
    return RTO_OK;
}

// The parts (or whole frames) of n <= kMaxBatch frames in one launch of k_trace_lean_triangles_batch; n launches when the
// lean triangle kernel is not the one in use.
static int launch_triangles_batch(rto_context* c, const rto_frame* frames, int n, const rto_partition* p, int shadow, float4* const* outs,
                                  hipStream_t s, bool shadeOut) {
    if (s != c->stream) c->otherStreams = true;
    const bool lean = c->d_triRec && c->canonical && c->numInternal > 0 && !c->culling &&
                      (c->kernelMode == RTO_KERNEL_AUTO || c->kernelMode == RTO_KERNEL_PACKED);
    if (!lean || n == 1) {
        for (int i = 0; i < n; i++) {
            const int rc = launch_triangles(c, &frames[i], p, shadow, outs[i], s, false, shadeOut);
            if (rc != RTO_OK) return rc;
        }
        return RTO_OK;
    }
    if (!c->d_triOffset) return fail(c, RTO_E_NO_OCTREE, "render_triangles: no leaf triangles uploaded");
    const bool capturing = stream_is_capturing(s);
    { const int rcOrder = order_after_cull(c, s); if (rcOrder != RTO_OK) return rcOrder; }
    const bool noEvents = capturing || c->eventsOff;
    RenderBatch B;
    B.n = n;
    int maxWaves = 0;
    rto_context::OrderState* st = nullptr;
    for (int i = 0; i < n; i++) {
        int rc = fill_params(c, &frames[i], p, B.P[i], s);
        if (rc != RTO_OK) return rc;
        B.out[i] = outs[i];
        if (B.P[i].tilesX * B.P[i].tilesY <= 0) { B.P[i].launchWaves = 0; continue; }
        const int solidRect[4] = { B.P[i].solidX0, B.P[i].solidY0, B.P[i].solidX1, B.P[i].solidY1 };
        if ((rc = prepare_schedule(c, s, capturing, true, false, 1, solidRect, B.P[i], &st, i)) != RTO_OK) return rc;
        maxWaves = std::max(maxWaves, B.P[i].launchWaves);
    }
    batch_mask_all_or_none(B);
    for (int i = 0; i < n; i++) {                                   // one box-relative order table per stream: see launch_trace_batch
        if (!st || B.P[i].launchWaves <= 0 || !B.P[i].tileCost) continue;
        const rto_context::OrderState::Table& T = st->tab[st->active];
        const bool fits = T.valid && B.P[i].boxW == T.box[2] && B.P[i].boxH == T.box[3] && B.P[i].traceWaves > 0;
        B.P[i].tileOrder = fits ? T.d : nullptr;
    }
    if (maxWaves <= 0) return RTO_OK;
    hipEvent_t evA = c->ev0, evB = c->ev1;
    if (!noEvents && c->ringUsed < c->ringStart.size()) { evA = c->ringStart[c->ringUsed]; evB = c->ringStop[c->ringUsed]; c->ringUsed++; }
    if (!noEvents) RTO_HIP(c, hipEventRecord(evA, s));
    LeanTriScene S{ c->d_triRec, c->d_tris };
#if defined(RTO_TRI_TIMELINE)
    S.timeline = nullptr;
#endif
    const size_t lds = (size_t)lean_wpb(1) * ((B.P[0].depth + 1) * kWave * sizeof(uint2) + kWave * sizeof(unsigned long long));
    for (int i = 0; i < n; i++) B.P[i].maskLdsBytes = (int)lds;
    const long long waves = (long long)maxWaves * n;
    const dim3 grid((unsigned)((waves + lean_wpb(1) - 1) / lean_wpb(1)) + (unsigned)(B.P[0].maskBlocks * n));
    if (c->maskMode == 2 && B.P[0].maskBlocks > 0)
        hipLaunchKernelGGL(k_trace_lean_triangles_batch<false>, dim3((unsigned)(B.P[0].maskBlocks * n)), dim3(lean_block(1)), lds, s, B, S, shadow);
    if (shadeOut) hipLaunchKernelGGL(k_trace_lean_triangles_batch<true>, grid, dim3(lean_block(1)), lds, s, B, S, shadow);
    else hipLaunchKernelGGL(k_trace_lean_triangles_batch<false>, grid, dim3(lean_block(1)), lds, s, B, S, shadow);
    RTO_HIP(c, hipGetLastError());
    if (!noEvents) RTO_HIP(c, hipEventRecord(evB, s));
    c->lastA = evA; c->lastB = evB;
    c->timed = !noEvents;
    return RTO_OK;
}

int rto_render_triangles_batch_device(rto_context* c, const rto_frame* frames, int n, const rto_partition* p, int shadow, int shade_payload,
                                      void* d_out, size_t frame_stride_bytes, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!frames || n < 1 || !d_out) return fail(c, RTO_E_INVALID, "rto_render_triangles_batch_device: NULL argument / empty batch");
    RTO_HIP(c, hipSetDevice(c->device));
    for (int i = 0; i < n; i++)
        if (frames[i].width != frames[0].width || frames[i].height != frames[0].height)
            return fail(c, RTO_E_INVALID, "rto_render_triangles_batch_device: the frames of a batch share width and height");
    for (int i0 = 0; i0 < n; i0 += kMaxBatch) {
        const int m = std::min(kMaxBatch, n - i0);
        float4* outs[kMaxBatch];
        for (int i = 0; i < m; i++) outs[i] = reinterpret_cast<float4*>(static_cast<char*>(d_out) + (size_t)(i0 + i) * frame_stride_bytes);
        const int rc = launch_triangles_batch(c, &frames[i0], m, p, shadow, outs, (hipStream_t)hip_stream, shade_payload != 0);
        if (rc != RTO_OK) return rc;
    }
    return RTO_OK;
}

int rto_render_triangles_device(rto_context* c, const rto_frame* f, const rto_partition* p, int shadow, void* d_out, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!d_out) return fail(c, RTO_E_INVALID, "rto_render_triangles_device: d_out is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    return launch_triangles(c, f, p, shadow, (float4*)d_out, (hipStream_t)hip_stream, false);
}

int rto_render_resident(rto_context* c, const rto_frame* f, int mode) {
    if (!c) return RTO_E_INVALID;
    if (!f) return fail(c, RTO_E_INVALID, "rto_render_resident: frame is NULL");
    if (mode < RTO_RESIDENT_OCTREE || mode > RTO_RESIDENT_TRIANGLES_SHADOW) return fail(c, RTO_E_INVALID, "rto_render_resident: unknown mode");
    RTO_HIP(c, hipSetDevice(c->device));
    if (f->width <= 0 || f->height <= 0) return fail(c, RTO_E_INVALID, "rto_render_resident: empty frame");
    const size_t pixels = (size_t)f->width * f->height;
    int rc = ensure_frame(c, pixels);
    if (rc != RTO_OK) return rc;
    c->residentW = c->residentH = 0;
    if (mode == RTO_RESIDENT_OCTREE) {
        RenderParams P;
        if ((rc = fill_params(c, f, nullptr, P)) != RTO_OK) return rc;
        if ((rc = launch_trace<kModeColor>(c, P, c->d_frame, c->stream)) != RTO_OK) return rc;
    } else {
        if ((rc = launch_triangles(c, f, nullptr, mode == RTO_RESIDENT_TRIANGLES_SHADOW ? 1 : 0, c->d_frame, c->stream, false)) != RTO_OK) return rc;
    }
    c->residentW = f->width; c->residentH = f->height;
    return RTO_OK;
}

int rto_resident_frame(rto_context* c, void** d_rgba, int* width, int* height) {
    if (!c) return RTO_E_INVALID;
    if (c->residentW <= 0) return fail(c, RTO_E_INVALID, "rto_resident_frame: nothing rendered with rto_render_resident yet");
    if (d_rgba) *d_rgba = c->d_frame;
    if (width) *width = c->residentW;
    if (height) *height = c->residentH;
    return RTO_OK;
}

int rto_download_resident(rto_context* c, float* host_rgba) {
    if (!c) return RTO_E_INVALID;
    if (!host_rgba) return fail(c, RTO_E_INVALID, "rto_download_resident: host_rgba is NULL");
    if (c->residentW <= 0) return fail(c, RTO_E_INVALID, "rto_download_resident: nothing rendered with rto_render_resident yet");
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipMemcpyAsync(host_rgba, c->d_frame, (size_t)c->residentW * c->residentH * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    return RTO_OK;
}

int rto_render_triangles_shade_device(rto_context* c, const rto_frame* f, const rto_partition* p, int shadow, void* d_shade, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!d_shade) return fail(c, RTO_E_INVALID, "rto_render_triangles_shade_device: d_shade is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    return launch_triangles(c, f, p, shadow, (float4*)d_shade, (hipStream_t)hip_stream, false, true);
}

int rto_render_triangles_host(rto_context* c, const rto_frame* f, int shadow, float* host_rgba, rto_stats* stats) {
    if (!c) return RTO_E_INVALID;
    if (!host_rgba || !f) return fail(c, RTO_E_INVALID, "rto_render_triangles_host: NULL argument");
    RTO_HIP(c, hipSetDevice(c->device));
    const size_t pixels = (size_t)f->width * f->height;
    int rc = ensure_frame(c, pixels);
    if (rc != RTO_OK) return rc;
    if (stats) RTO_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    if ((rc = launch_triangles(c, f, nullptr, shadow, c->d_frame, c->stream, stats != nullptr)) != RTO_OK) return rc;
    Counters h{ 0, 0, 0 };
    if (stats) RTO_HIP(c, hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipMemcpyAsync(host_rgba, c->d_frame, pixels * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    if (stats) { stats->rays = pixels; stats->pops = h.pops; stats->hits = h.hits; stats->capped = 0; }
    return RTO_OK;
}

int rto_octree_ray_skip(rto_context* c, const float ro[3], const float* rd, int64_t n, float t_min, float t_max,
                        int use_visibility, float* out_t) {
    if (!c) return RTO_E_INVALID;
    if (!ro || !rd || !out_t || n < 0) return fail(c, RTO_E_INVALID, "rto_octree_ray_skip: NULL argument");
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_octree_ray_skip: no octree uploaded");
    if (n == 0) return RTO_OK;
    RTO_HIP(c, hipSetDevice(c->device));
    BuildScratch scratch(c->stream);                 // stream-ordered pool: no hipMalloc / hipFree per call
    float *d_rd = nullptr, *d_out = nullptr;
    RTO_HIP(c, scratch.alloc(&d_rd, (size_t)n * 3));
    RTO_HIP(c, scratch.alloc(&d_out, (size_t)n));
    RTO_HIP(c, hipMemcpyAsync(d_rd, rd, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const uint8_t* vis = (use_visibility && c->culling) ? c->d_vis : nullptr;
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    // canonical trees: the descriptor form (latency of a ray = one 8-byte load per visited internal node);
    // anything else, or RTO_KERNEL_GENERIC: the 60-byte node form
    const size_t ldsSkip = (size_t)(kBlock / kWave) * std::max(c->depth, 1) * kWave * sizeof(uint4);      // the rays' frames, [wave][level][lane]
    if (c->canonical && c->numInternal > 0 && c->kernelMode != RTO_KERNEL_GENERIC && ldsSkip <= 62 * 1024)
        hipLaunchKernelGGL(k_octree_ray_skip_packed, dim3(blocks), dim3(kBlock), ldsSkip, c->stream, c->d_desc, vis, vis ? 1 : 0, c->rootSize, std::max(c->depth, 1),
                           c->gridMin[0], c->gridMin[1], c->gridMin[2], c->voxelSize, ro[0], ro[1], ro[2], d_rd, n, t_min, t_max, d_out);
    else
        hipLaunchKernelGGL(k_octree_ray_skip, dim3(blocks), dim3(kBlock), 0, c->stream, c->d_nodes, vis,
                           c->gridMin[0], c->gridMin[1], c->gridMin[2], c->voxelSize, ro[0], ro[1], ro[2],
                           d_rd, n, t_min, t_max, d_out);
    RTO_HIP(c, hipGetLastError());
    RTO_HIP(c, hipMemcpyAsync(out_t, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    return RTO_OK;
}

// ---- the reference's closest-hit traversal (its earlier, block-commented shader: S/RT:46-166) --------------------------------
static int launch_closest(rto_context* c, const rto_frame* f, const rto_partition* p, float4* d_out, hipStream_t s, bool count) {
    if (s != c->stream) c->otherStreams = true;
    RenderParams P;
    int rc = fill_params(c, f, p, P, s);
    if (rc != RTO_OK) return rc;
    if ((rc = order_after_cull(c, s)) != RTO_OK) return rc;
    const int tiles = P.tilesX * P.tilesY;
    if (tiles <= 0) return RTO_OK;
    // canonical trees without a frustum update in force: the descriptor form (k_closest_lean); RTO_KERNEL_GENERIC keeps the node-by-node form
    if (c->canonical && c->numInternal > 0 && !c->culling && c->kernelMode != RTO_KERNEL_GENERIC) {
        const int blocksL = (tiles + (kBlock / kWave) - 1) / (kBlock / kWave);
        const size_t lds = (size_t)(kBlock / kWave) * (P.depth + 1) * kWave * sizeof(uint2);
        if (count) hipLaunchKernelGGL(k_closest_lean<kModeSteps>, dim3(blocksL), dim3(kBlock), lds, s, P, c->d_desc, d_out, c->d_counters);
        else if (c->kernelMode == RTO_KERNEL_PACKED_V1) hipLaunchKernelGGL(k_closest_lean<kModeColor>, dim3(blocksL), dim3(kBlock), lds, s, P, c->d_desc, d_out, c->d_counters);
        else {
            // launch geometry, launch order and occupancy mask of the first-hit frames (path 3: tables of its own kind on the stream)
            rto_context::OrderState* st = nullptr;
            const int solidRect[4] = { P.solidX0, P.solidY0, P.solidX1, P.solidY1 };
            if ((rc = prepare_schedule(c, s, stream_is_capturing(s), true, false, 3, solidRect, P, &st, 0, true)) != RTO_OK) return rc;
            P.maskLdsBytes = (int)lds;
            const int blocksN = (P.launchWaves + (kBlock / kWave) - 1) / (kBlock / kWave) + P.maskBlocks;
            if (c->maskMode == 2 && P.maskBlocks > 0) hipLaunchKernelGGL(k_closest_near_first, dim3(P.maskBlocks), dim3(kBlock), lds, s, P, c->d_desc, d_out);
            hipLaunchKernelGGL(k_closest_near_first, dim3(blocksN), dim3(kBlock), lds, s, P, c->d_desc, d_out);
        }
        RTO_HIP(c, hipGetLastError());
        return RTO_OK;
    }
    // with a frustum update in force the traversal runs over the compacted array, as the reference's culled render does (S/RT:765-812)
    if ((rc = ensure_compact(c, s)) != RTO_OK) return rc;
    const rto_node* nodes = c->culling ? c->d_compact : c->d_nodes;
    if (c->culling) P.rootVisible = c->visibleNodes > 0 ? 1 : 0;
    const int blocks = (tiles + (kBlock / kWave) - 1) / (kBlock / kWave);
    if (count) hipLaunchKernelGGL(k_trace_closest<kModeSteps>, dim3(blocks), dim3(kBlock), 0, s, P, nodes, d_out, c->d_counters);
    else hipLaunchKernelGGL(k_trace_closest<kModeColor>, dim3(blocks), dim3(kBlock), 0, s, P, nodes, d_out, c->d_counters);
    RTO_HIP(c, hipGetLastError());
    return RTO_OK;
}

int rto_render_closest_device(rto_context* c, const rto_frame* f, const rto_partition* p, void* d_out, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    if (!d_out) return fail(c, RTO_E_INVALID, "rto_render_closest_device: d_out is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    return launch_closest(c, f, p, (float4*)d_out, (hipStream_t)hip_stream, false);
}

int rto_render_closest_host(rto_context* c, const rto_frame* f, float* host_rgba, rto_stats* stats) {
    if (!c) return RTO_E_INVALID;
    if (!f || !host_rgba) return fail(c, RTO_E_INVALID, "rto_render_closest_host: NULL argument");
    RTO_HIP(c, hipSetDevice(c->device));
    if (f->width <= 0 || f->height <= 0) return fail(c, RTO_E_INVALID, "render: width/height must be positive");
    const size_t pixels = (size_t)f->width * f->height;
    int rc = ensure_frame(c, pixels);
    if (rc != RTO_OK) return rc;
    if (stats) RTO_HIP(c, hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));
    if ((rc = launch_closest(c, f, nullptr, c->d_frame, c->stream, stats != nullptr)) != RTO_OK) return rc;
    Counters h{ 0, 0, 0 };
    if (stats) RTO_HIP(c, hipMemcpyAsync(&h, c->d_counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipMemcpyAsync(host_rgba, c->d_frame, pixels * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    if (stats) { stats->rays = pixels; stats->pops = h.pops; stats->hits = h.hits; stats->capped = 0; }
    return RTO_OK;
}

// ---- N1 as a render mode + N1's consumer ------------------------------------------------------------------------------
static int launch_skip_render(rto_context* c, const rto_frame* f, const rto_partition* p, int use_visibility, float4* d_rgba, float* d_dist, hipStream_t s) {
    if (!d_rgba && !d_dist) return fail(c, RTO_E_INVALID, "rto_render_skip: neither output given");
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_render_skip: no octree uploaded");
    if (!(c->canonical && c->numInternal > 0)) return fail(c, RTO_E_UNSUPPORTED, "rto_render_skip: needs a canonical BFS octree (what setOctree / rto_build_octree produce)");
    if (s != c->stream) c->otherStreams = true;
    RenderParams P;
    int rc = fill_params(c, f, p, P, s);
    if (rc != RTO_OK) return rc;
    if ((rc = order_after_cull(c, s)) != RTO_OK) return rc;
    const int tiles = P.tilesX * P.tilesY;
    if (tiles <= 0) return RTO_OK;
    const size_t lds = (size_t)(kBlock / kWave) * P.depth * kWave * sizeof(uint4);
    if (lds > 64 * 1024) return fail(c, RTO_E_UNSUPPORTED, "rto_render_skip: octree too deep for the LDS frames");
    const uint8_t* vis = (use_visibility && c->culling) ? c->d_vis : nullptr;
    // launch geometry, launch order and occupancy mask of the first-hit frames (path 2: tables of its own kind on the stream)
    rto_context::OrderState* st = nullptr;
    const int solidRect[4] = { P.solidX0, P.solidY0, P.solidX1, P.solidY1 };
    if ((rc = prepare_schedule(c, s, stream_is_capturing(s), true, false, 2, solidRect, P, &st, 0, true)) != RTO_OK) return rc;
    P.maskLdsBytes = (int)lds;
    const int blocks = (P.launchWaves + (kBlock / kWave) - 1) / (kBlock / kWave) + P.maskBlocks;
    if (c->maskMode == 2 && P.maskBlocks > 0)
        hipLaunchKernelGGL(k_skip_render, dim3(P.maskBlocks), dim3(kBlock), lds, s, P, c->d_desc, vis, vis ? 1 : 0, d_rgba, d_dist);
    hipLaunchKernelGGL(k_skip_render, dim3(blocks), dim3(kBlock), lds, s, P, c->d_desc, vis, vis ? 1 : 0, d_rgba, d_dist);
    RTO_HIP(c, hipGetLastError());
    return RTO_OK;
}

int rto_render_skip_device(rto_context* c, const rto_frame* f, const rto_partition* p, int use_visibility, void* d_rgba, void* d_dist, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    return launch_skip_render(c, f, p, use_visibility, (float4*)d_rgba, (float*)d_dist, (hipStream_t)hip_stream);
}

int rto_render_skip_host(rto_context* c, const rto_frame* f, int use_visibility, float* host_rgba, float* host_dist) {
    if (!c) return RTO_E_INVALID;
    if (!f || (!host_rgba && !host_dist)) return fail(c, RTO_E_INVALID, "rto_render_skip_host: NULL argument");
    RTO_HIP(c, hipSetDevice(c->device));
    if (f->width <= 0 || f->height <= 0) return fail(c, RTO_E_INVALID, "render: width/height must be positive");
    const size_t pixels = (size_t)f->width * f->height;
    int rc = ensure_frame(c, pixels + (pixels + 3) / 4);           // RGBA frame + the distances behind it
    if (rc != RTO_OK) return rc;
    float* d_dist = reinterpret_cast<float*>(c->d_frame + pixels);
    if ((rc = launch_skip_render(c, f, nullptr, use_visibility, host_rgba ? c->d_frame : nullptr, host_dist ? d_dist : nullptr, c->stream)) != RTO_OK) return rc;
    if (host_rgba) RTO_HIP(c, hipMemcpyAsync(host_rgba, c->d_frame, pixels * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (host_dist) RTO_HIP(c, hipMemcpyAsync(host_dist, d_dist, pixels * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    return RTO_OK;
}

static int launch_probe_skip(rto_context* c, const float view[16], const float cam_pos[3], float aspect, int use_visibility, float* d_skip, float* d_probeT, hipStream_t s) {
    if (!view || !cam_pos || !d_skip) return fail(c, RTO_E_INVALID, "rto_probe_skip: NULL argument");
    if (c->numNodes <= 0) return fail(c, RTO_E_NO_OCTREE, "rto_probe_skip: no octree uploaded");
    if (!(c->canonical && c->numInternal > 0)) return fail(c, RTO_E_UNSUPPORTED, "rto_probe_skip: needs a canonical BFS octree");
    if (s != c->stream) c->otherStreams = true;
    { const int rcOrder = order_after_cull(c, s); if (rcOrder != RTO_OK) return rcOrder; }
    ProbeParams Q;
    // S/VR:1608-1612: P = perspective(radians(45), aspect, 0.1, 5000); invV = inverse(V); invP = inverse(P) -- pixel independent, on the host
    const rtmath::mat4 P = rtmath::perspective(rtmath::radians(45.0f), aspect, 0.1f, 5000.0f);
    const rtmath::mat4 invP = rtmath::inverse(P), invV = rtmath::inverse(rtmath::mat4::from(view));
    std::memcpy(Q.invP, invP.data(), sizeof Q.invP);
    std::memcpy(Q.invV, invV.data(), sizeof Q.invV);
    std::memcpy(Q.eye, cam_pos, sizeof Q.eye);
    Q.gx = c->gridMin[0]; Q.gy = c->gridMin[1]; Q.gz = c->gridMin[2]; Q.vs = c->voxelSize;
    Q.rootSize = c->rootSize; Q.depth = c->depth > 0 ? c->depth : 1;
    const size_t lds = (size_t)Q.depth * kWave * sizeof(uint4);
    if (lds > 64 * 1024) return fail(c, RTO_E_UNSUPPORTED, "rto_probe_skip: octree too deep for the LDS frames");
    const uint8_t* vis = (use_visibility && c->culling) ? c->d_vis : nullptr;
    hipLaunchKernelGGL(k_probe_skip, dim3(1), dim3(kWave), lds, s, Q, c->d_desc, vis, vis ? 1 : 0, d_skip, d_probeT);
    RTO_HIP(c, hipGetLastError());
    return RTO_OK;
}

int rto_probe_skip_device(rto_context* c, const float view[16], const float cam_pos[3], float aspect, int use_visibility, void* d_skip, void* hip_stream) {
    if (!c) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    return launch_probe_skip(c, view, cam_pos, aspect, use_visibility, (float*)d_skip, nullptr, (hipStream_t)hip_stream);
}

int rto_probe_skip_host(rto_context* c, const float view[16], const float cam_pos[3], float aspect, int use_visibility, float* io_skip, float* probe_t) {
    if (!c) return RTO_E_INVALID;
    if (!io_skip) return fail(c, RTO_E_INVALID, "rto_probe_skip_host: io_skip is NULL");
    RTO_HIP(c, hipSetDevice(c->device));
    BuildScratch scratch(c->stream);
    float* d = nullptr;
    RTO_HIP(c, scratch.alloc(&d, 64));
    RTO_HIP(c, hipMemcpyAsync(d, io_skip, sizeof(float), hipMemcpyHostToDevice, c->stream));
    const int rc = launch_probe_skip(c, view, cam_pos, aspect, use_visibility, d, probe_t ? d + 8 : nullptr, c->stream);
    if (rc != RTO_OK) return rc;
    RTO_HIP(c, hipMemcpyAsync(io_skip, d, sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (probe_t) RTO_HIP(c, hipMemcpyAsync(probe_t, d + 8, 49 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    RTO_HIP(c, hipStreamSynchronize(c->stream));
    return RTO_OK;
}

int rto_last_kernel_ms(rto_context* c, float* ms) {
    if (!c || !ms) return RTO_E_INVALID;
    if (!c->timed) return fail(c, RTO_E_INVALID, "rto_last_kernel_ms: no kernel launched yet");
    RTO_HIP(c, hipSetDevice(c->device));
    hipEvent_t a = c->lastA ? c->lastA : c->ev0, b = c->lastB ? c->lastB : c->ev1;
    RTO_HIP(c, hipEventSynchronize(b));
    RTO_HIP(c, hipEventElapsedTime(ms, a, b));
    return RTO_OK;
}

int rto_timing_begin(rto_context* c, int capacity) {
    if (!c || capacity < -1) return RTO_E_INVALID;
    c->eventsOff = capacity < 0;
    if (capacity < 0) capacity = 0;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    for (hipEvent_t e : c->ringStart) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ringStop) (void)hipEventDestroy(e);
    c->ringStart.clear(); c->ringStop.clear(); c->ringUsed = 0; c->lastA = c->lastB = nullptr;
    for (int i = 0; i < capacity; i++) {
        hipEvent_t a, b;
        RTO_HIP(c, hipEventCreate(&a)); c->ringStart.push_back(a);
        RTO_HIP(c, hipEventCreate(&b)); c->ringStop.push_back(b);
    }
    return RTO_OK;
}

int rto_timing_read(rto_context* c, float* ms, int capacity, int* count) {
    if (!c || !count) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    *count = (int)c->ringUsed;
    if (!ms) return RTO_OK;
    if (capacity < (int)c->ringUsed) return fail(c, RTO_E_INVALID, "rto_timing_read: capacity too small");
    for (size_t i = 0; i < c->ringUsed; i++) RTO_HIP(c, hipEventElapsedTime(&ms[i], c->ringStart[i], c->ringStop[i]));
    return RTO_OK;
}

void* rto_stream(rto_context* c) { return c ? (void*)c->stream : nullptr; }

int rto_synchronize(rto_context* c) {
    if (!c) return RTO_E_INVALID;
    RTO_HIP(c, hipSetDevice(c->device));
    RTO_HIP(c, hipDeviceSynchronize());
    return RTO_OK;
}

}  // extern "C"

#include "rto_split.inc"
#include "rto_comm.inc"
