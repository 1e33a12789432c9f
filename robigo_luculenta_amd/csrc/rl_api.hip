// rl_api.hip -- implementation of include/robigo_luculenta.h over the gfx950 kernels.
//
// Ordering model: every trace, plot and gather unit owns a NON-blocking HIP stream and nothing relies on the
// null stream's implicit ordering (RCCL and a host framework's side streams are non-blocking too).  Every
// hand-over between units is an event:
//   * a fused render writes the plot unit's buffer from the trace unit's stream: the plot stream waits for it
//     (so "everything written into this plot unit so far" is always the tail of the plot unit's own stream),
//     and the render itself waits for the plot unit's last clear;
//   * PlotUnit::plot reads mapped_photons on the plot stream after the trace unit's render; the trace unit's
//     next render waits for that plot;
//   * GatherUnit::accumulate runs on the gather stream after the plot stream's tail, and the plot stream
//     continues after the clear; tonemap kernels run on the gather stream;
//   * the GatherUnit-time exchange (rl_plot_unit_reduce, RCCL) is enqueued on the plot stream.
//   * the blocking render calls (open launches, see `Session` below) are ordered on the HOST instead: before a call
//     is appended to a kernel that is already running, the host waits for whatever still reads or clears its target,
//     and the call (or whoever ends a begun one) returns only when the kernel has reported its last path.
// Downloads synchronise the stream that produced the data.  Trace and plot work of different units overlap on
// the device.  Entry points never throw.
#include <hip/hip_runtime.h>
#include <sys/prctl.h>

#include <dlfcn.h>
#include <unistd.h> // fsync

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h> // types and prototypes only: the library is bound at run time (rccl_api below)

#include "../../include/robigo_luculenta.h"
#include "../../include/robigo_luculenta_debug.h"
#include "rl_kernels.hip.h"
#include "rl_scene.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

} // namespace

// Internal (not in the C header): lets rl_app.cpp hand a worker thread's error message to the thread that
// called rl_app_run -- rl_last_error() is per thread.
void rl_internal_set_last_error(const std::string& msg) { g_error = msg; }

namespace {

#define RL_HIP(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(RL_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                       \
    } while (0)

int use_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RL_E_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= n) return fail(RL_E_INVALID, "device index out of range");
    RL_HIP(hipSetDevice(device));
    return RL_OK;
}

struct EventPair {
    hipEvent_t start, stop;
};

} // namespace

struct RlScene {
    int device;
    RlF4* blob;
    RlSceneLayout lay;
    size_t staged_bytes; // the whole blob
    size_t tables_bytes; // its tables: records [off_planes, off_objects)
};

namespace {
struct Session;
enum { RL_SESSIONS_PINNED = 1000 }; // internal: session_begin found every open-launch slot held by begun renders

// Paths and segments of a trace unit's calls that open launches served.  Shared with the tickets of the fused renders
// begun on the unit, which are ended on other threads (whoever uses the plot unit next) and possibly after the unit is gone.
struct UnitCounters {
    std::atomic<uint64_t> paths{0}, segments{0};
    std::atomic<uint64_t> kernel_ns{0}; // the unit's share (by paths) of the open launches that carried its calls
};

// A blocking render that was begun and not yet ended.  The ticket of an un-fused render lives in its trace unit (whose
// photons it fills), that of a fused render in the PLOT unit it splats into: the trace unit only lends its image size
// and fetch mode and is free for the next call -- on whatever thread -- at once, so the ticket holds nothing of it but
// the counters to credit.
struct Ticket {
    bool pending = false;
    Session* session = nullptr; // the open launch the call was appended to; null: a plain launch (ragged or huge batch)
    uint32_t job = 0;
    uint64_t paths = 0;
    int device = 0;
    std::shared_ptr<UnitCounters> counters;
    double presync_us = 0.0, admit_us = 0.0;
};
}

struct RlTraceUnit {
    int device;
    uint32_t id, width, height, n_photons;
    RlMappedPhoton* photons;
    unsigned long long* queue; // 3 counters, see rl_trace_kernel
    hipStream_t stream;
    hipEvent_t rendered; // recorded on `stream` after everything that fills this unit's photons
    int fetch;
    int cu_count;
    std::vector<EventPair> pending, pool;
    double kernel_ms;
    uint64_t launches;
    size_t tuned_dyn;  // launch configuration last set up for this unit: dynamic LDS bytes,
    const void* tuned_kernel; // kernel instantiation,
    int tuned_per_cu;  // resident workgroups per CU (0 = not set up yet)
    std::shared_ptr<UnitCounters> counters = std::make_shared<UnitCounters>(); // of this unit's calls that open launches served
    Ticket ticket;                                                               // rl_trace_unit_render_begin
};

struct RlPlotUnit {
    int device;
    uint32_t id, width, height;
    float* xyz;
    bool owns;
    RlF4* cie;
    hipStream_t stream; // rl_plot_kernel launches, clears, the RCCL exchange; its tail = every write into xyz so far
    hipEvent_t plotted; // recorded after the last plot kernel of a PlotUnit::plot call
    hipEvent_t ready;   // recorded on `stream` when a gather (or a reader on another stream) takes the buffer
    hipEvent_t cleared; // recorded on the gather stream after accumulate + clear
    hipEvent_t tail;    // scratch: recorded on `stream` by a fused launch that has to start behind everything queued there
    std::vector<EventPair> exchanges, exchange_pool; // around every rl_plot_unit_reduce on `stream`, not yet read (rl_plot_unit_exchange_stats); spare pairs
    uint64_t exchange_count = 0;
    double exchange_ms = 0.0;
    Ticket ticket;      // rl_trace_unit_render_fused_begin: ended by whatever uses the buffer next (plot_settle)
};

struct RlGatherUnit {
    int device;
    uint32_t width, height;
    float* acc;
    float* comp;
    hipStream_t stream; // accumulate and tonemap kernels
};

struct RlTonemapUnit {
    int device;
    uint32_t width, height;
    uint8_t* rgb;
    float* srgb;
    float* max_intensity;
    hipStream_t last_stream; // the gather stream the last tonemap ran on (downloads wait for it)
};

// One rank of an RCCL communicator (rl_comm_*).
struct RlComm {
    int device, rank, world;
    void* nccl; // ncclComm_t
};

namespace {

int drain_events(RlTraceUnit* u) {
    for (EventPair& ep : u->pending) {
        RL_HIP(hipEventSynchronize(ep.stop));
        float ms = 0.0f;
        RL_HIP(hipEventElapsedTime(&ms, ep.start, ep.stop));
        u->kernel_ms += (double)ms;
        u->pool.push_back(ep);
    }
    u->pending.clear();
    return RL_OK;
}

// The instantiation of rl_trace_kernel for a launch: primitives staged in LDS or fetched from global memory, fused with
// the splat or not, an open launch or a plain one, prisms with a second bound or without.
typedef void (*TraceKernel)(const RlF4*, RlSceneLayout, RlTraceJob, RlMappedPhoton*, float*, unsigned long long*, const RlJobEntry*,
                            RlOpenDev*, RlOpenCtl*);
std::atomic<uint64_t> g_variant_launches[24]; // rl_debug_variant_launches: launches per instantiation since the library was loaded
// stage: RL_STAGE_NONE / RL_STAGE_TABLES / RL_STAGE_ALL.  Index = the four template arguments as bits -- 8: the whole scene in LDS,
// 4: fused, 2: open, 1: cylinders -- and 16 + the low three for the variants that stage the tables only.
TraceKernel trace_kernel_variant(int stage, bool fused, bool open, bool cyl) {
    const int low = (fused ? 4 : 0) | (open ? 2 : 0) | (cyl ? 1 : 0);
    const int index = stage == RL_STAGE_TABLES ? 16 + low : (stage == RL_STAGE_ALL ? 8 : 0) | low;
    g_variant_launches[index].fetch_add(1, std::memory_order_relaxed);
    static const TraceKernel table[24] = {
        rl_trace_kernel<RL_STAGE_NONE, false, false>, rl_trace_kernel<RL_STAGE_NONE, false, true>, rl_trace_kernel_open<RL_STAGE_NONE, false, false>, rl_trace_kernel_open<RL_STAGE_NONE, false, true>,
        rl_trace_kernel<RL_STAGE_NONE, true, false>, rl_trace_kernel<RL_STAGE_NONE, true, true>, rl_trace_kernel_open<RL_STAGE_NONE, true, false>, rl_trace_kernel_open<RL_STAGE_NONE, true, true>,
        rl_trace_kernel<RL_STAGE_ALL, false, false>, rl_trace_kernel<RL_STAGE_ALL, false, true>, rl_trace_kernel_open<RL_STAGE_ALL, false, false>, rl_trace_kernel_open<RL_STAGE_ALL, false, true>,
        rl_trace_kernel<RL_STAGE_ALL, true, false>, rl_trace_kernel<RL_STAGE_ALL, true, true>, rl_trace_kernel_open<RL_STAGE_ALL, true, false>, rl_trace_kernel_open<RL_STAGE_ALL, true, true>,
        rl_trace_kernel<RL_STAGE_TABLES, false, false>, rl_trace_kernel<RL_STAGE_TABLES, false, true>, rl_trace_kernel_open<RL_STAGE_TABLES, false, false>, rl_trace_kernel_open<RL_STAGE_TABLES, false, true>,
        rl_trace_kernel<RL_STAGE_TABLES, true, false>, rl_trace_kernel<RL_STAGE_TABLES, true, true>, rl_trace_kernel_open<RL_STAGE_TABLES, true, false>, rl_trace_kernel_open<RL_STAGE_TABLES, true, true>,
    };
    return table[index];
}
// Ring T of rl_scan_wave: 512 bytes per wave in front of the scratch blocks, for scenes whose cull table has a third level.
size_t ring_t_bytes(const RlScene* scene) { return scene->lay.n_cluster_supers != 0u ? (size_t)(RL_TRACE_BLOCK / 64) * 512u : 0u; }
// What a launch stages in LDS beside `scratch_bytes` of per-wave scratch: the whole scene where it fits, its tables where
// those do, nothing otherwise (or when the unit was created with RL_FETCH_GLOBAL); *bytes = the staged size.
int stage_of(const RlScene* scene, int fetch, size_t scratch_bytes, size_t* bytes) {
    *bytes = 0;
    if (fetch != RL_FETCH_LDS) return RL_STAGE_NONE;
    // (what is staged is rounded up to 512 bytes: the waves' scratch behind it is 512-byte aligned, rl_trace_body)
    const size_t all = (scene->staged_bytes + 511) & ~(size_t)511, tables = (scene->tables_bytes + 511) & ~(size_t)511;
    if (scene->lay.n_cluster_supers == 0u && all + scratch_bytes <= 160 * 1024) { // (a table with a third level: the whole-scene variants do not carry its code)
        *bytes = all;
        return RL_STAGE_ALL;
    }
    if (tables + scratch_bytes <= 160 * 1024) {
        *bytes = tables;
        return RL_STAGE_TABLES;
    }
    return RL_STAGE_NONE;
}

// One launch of the trace kernel on u's stream: n_paths paths from first_path on, into `photons` (un-fused) or splatted
// into plot_unit's buffer (fused).
int launch_trace(RlTraceUnit* u, const RlScene* scene, RlMappedPhoton* photons, RlPlotUnit* plot_unit, uint64_t seed,
                 uint32_t stream_id, uint64_t first_path, uint64_t n_paths) {
    if (n_paths == 0) return RL_OK;
    if (first_path + n_paths < first_path || first_path + n_paths == ~0ull)
        return fail(RL_E_INVALID, "path indices must stay below 2^64 - 1");
    float* plot = plot_unit ? plot_unit->xyz : nullptr;
    if (scene->device != u->device) return fail(RL_E_STATE, "scene and trace unit live on different devices");
    RlTraceJob job;
    job.width = u->width;
    job.height = u->height;
    job.aspect_ratio = (float)u->width / (float)u->height; // trace_unit.rs:73
    job.stream = stream_id;
    job.seed = seed;
    job.first_path = first_path;
    job.n_paths = n_paths;
    job.grace_ticks = 0;
    job.reserved = 0;
    job.wm1 = (float)(int)u->width - 1.0f;
    job.hm1 = (float)(int)u->height - 1.0f;

    // One workgroup of RL_TRACE_BLOCK threads per CU: [scene blob][per-wave scratch] in dynamic LDS.
    const size_t scratch_bytes = (RL_TRACE_BLOCK / 64) * sizeof(RlWaveScratch) + ring_t_bytes(scene);
    size_t blob_bytes = 0;
    const int stage = stage_of(scene, u->fetch, scratch_bytes, &blob_bytes);
    const bool fused = plot != nullptr;
    const bool cyl = scene->lay.prism_cylinders != 0u;
    auto kernel = trace_kernel_variant(stage, fused, false, cyl);
    const size_t dyn = scratch_bytes + blob_bytes;
    if (u->tuned_per_cu == 0 || u->tuned_dyn != dyn || u->tuned_kernel != (const void*)kernel) { // once per (unit, scene size, variant)
        // The limit is a property of the function, shared by every unit: always raise it to the whole LDS.
        RL_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int per_cu = 1;
        RL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RL_TRACE_BLOCK, dyn));
        u->tuned_per_cu = per_cu < 1 ? 1 : per_cu;
        if (getenv("RL_DEBUG_LAUNCH")) fprintf(stderr, "rl: trace launch: %d workgroup(s) of %d threads per CU, %zu bytes of LDS each, stage %d\n", per_cu, RL_TRACE_BLOCK, dyn, stage);
        u->tuned_dyn = dyn;
        u->tuned_kernel = (const void*)kernel; // (the instantiation itself, not some of its template arguments -- `cyl` was not among them)
    }
    uint64_t blocks = (uint64_t)u->cu_count * (uint64_t)u->tuned_per_cu;
    const uint64_t needed = (n_paths + RL_TRACE_BLOCK - 1) / RL_TRACE_BLOCK;
    if (blocks > needed) blocks = needed;

    EventPair ep;
    if (!u->pool.empty()) {
        ep = u->pool.back();
        u->pool.pop_back();
    } else {
        RL_HIP(hipEventCreate(&ep.start));
        RL_HIP(hipEventCreate(&ep.stop));
    }
    if (plot_unit) {
        RL_HIP(hipStreamWaitEvent(u->stream, plot_unit->cleared, 0)); // the splat must not race the last gather's clear
        // nor anything queued on the plot unit's own stream since (rl_plot_unit_add, the RCCL reduce, an upload)
        RL_HIP(hipEventRecord(plot_unit->tail, plot_unit->stream));
        RL_HIP(hipStreamWaitEvent(u->stream, plot_unit->tail, 0));
    }
    RL_HIP(hipMemsetAsync(u->queue, 0, sizeof(unsigned long long), u->stream));
    RL_HIP(hipEventRecord(ep.start, u->stream));
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(RL_TRACE_BLOCK), dyn, u->stream, scene->blob, scene->lay, job, photons,
                       plot, u->queue, (const RlJobEntry*)nullptr, (RlOpenDev*)nullptr, (RlOpenCtl*)nullptr);
    RL_HIP(hipGetLastError());
    RL_HIP(hipEventRecord(ep.stop, u->stream));
    RL_HIP(hipEventRecord(u->rendered, u->stream));
    if (plot_unit) RL_HIP(hipStreamWaitEvent(plot_unit->stream, u->rendered, 0)); // the plot stream's tail covers this splat
    u->pending.push_back(ep);
    u->launches += 1;
    if (u->pending.size() > 512) return drain_events(u);
    return RL_OK;
}

unsigned grid_for(uint64_t work_items, int cu_count) {
    uint64_t blocks = (work_items + RL_BLOCK - 1) / RL_BLOCK;
    const uint64_t cap = (uint64_t)cu_count * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int cu_count_of(int device, int* out) {
    // hipGetDeviceProperties is slow; the answer never changes, so remember it per device.
    static std::mutex lock;
    static int cached[64];
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> guard(lock);
        if (cached[device] > 0) {
            *out = cached[device];
            return RL_OK;
        }
    }
    hipDeviceProp_t prop;
    RL_HIP(hipGetDeviceProperties(&prop, device));
    *out = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> guard(lock);
        cached[device] = *out;
    }
    return RL_OK;
}

} // namespace

extern "C" {

const char* rl_last_error(void) { return g_error.c_str(); }

int rl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rl_device_pci_bus_id(int device, char* out, uint32_t cap) {
    if (!out || cap < 16) return fail(RL_E_INVALID, "output buffer too small for a PCI bus id");
    const int rc = use_device(device);
    if (rc != RL_OK) return rc;
    RL_HIP(hipDeviceGetPCIBusId(out, (int)cap, device));
    return RL_OK;
}

const char* rl_version(void) { return "robigo-luculenta_amd 0.3 (gfx950)"; }

#ifndef RL_BUILD_ID
#define RL_BUILD_ID "unknown"
#endif
const char* rl_build_id(void) { return RL_BUILD_ID; }

// ---- scene --------------------------------------------------------------------------------------

int rl_scene_builtin_desc(int which, int param, RlObjectDesc* objects, uint32_t cap, uint32_t* n_objects,
                          RlCameraDesc* camera) {
    std::vector<RlObjectDesc> v;
    RlCameraDesc cam;
    const uint32_t n = rl_builtin_scene(which, param, &v, &cam);
    if (n == 0) return fail(RL_E_INVALID, "unknown built-in scene");
    if (n_objects) *n_objects = n;
    if (camera) *camera = cam;
    if (!objects || cap < n) return fail(RL_E_INVALID, "object array too small for the built-in scene");
    std::memcpy(objects, v.data(), n * sizeof(RlObjectDesc));
    return RL_OK;
}

int rl_scene_desc_save(const char* path, const RlSceneDesc* desc) {
    if (!path || !desc || (!desc->objects && desc->n_objects)) return fail(RL_E_INVALID, "null argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RL_E_IO, std::string("failed to open file ") + path);
    const uint32_t header[2] = {1u, desc->n_objects};
    bool ok = std::fwrite("RLSC", 1, 4, f) == 4 && std::fwrite(header, 4, 2, f) == 2 &&
              std::fwrite(&desc->camera, sizeof(RlCameraDesc), 1, f) == 1 &&
              std::fwrite(desc->objects, sizeof(RlObjectDesc), desc->n_objects, f) == desc->n_objects;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RL_OK : fail(RL_E_IO, std::string("failed to write scene file ") + path);
}

int rl_scene_desc_load(const char* path, RlObjectDesc* objects, uint32_t cap, uint32_t* n_objects, RlCameraDesc* camera) {
    if (!path) return fail(RL_E_INVALID, "null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(RL_E_IO, std::string("failed to open file ") + path);
    char magic[4];
    uint32_t header[2] = {0, 0};
    RlCameraDesc cam;
    bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "RLSC", 4) == 0 && std::fread(header, 4, 2, f) == 2 &&
              header[0] == 1u && std::fread(&cam, sizeof cam, 1, f) == 1;
    if (!ok) {
        std::fclose(f);
        return fail(RL_E_IO, std::string("not a version-1 RLSC scene file: ") + path);
    }
    if (n_objects) *n_objects = header[1];
    if (camera) *camera = cam;
    if (!objects || cap < header[1]) {
        std::fclose(f);
        return fail(RL_E_INVALID, "object array too small for the scene file");
    }
    ok = std::fread(objects, sizeof(RlObjectDesc), header[1], f) == header[1];
    std::fclose(f);
    return ok ? RL_OK : fail(RL_E_IO, std::string("truncated scene file ") + path);
}

int rl_scene_create(const RlSceneDesc* desc, int device, RlScene** out) {
    if (!out) return fail(RL_E_INVALID, "null output handle");
    *out = nullptr;
    RlFlatScene fs;
    const char* err = "";
    int rc = rl_flatten_scene(desc, &fs, &err);
    if (rc != RL_OK) return fail(rc, err);
    if ((rc = use_device(device)) != RL_OK) return rc;

    // One blob: spheres || planes | parabs | prisms | cull table | prism cylinders | camera || objects | cie | sphere_obj | sphere_r2 (padded to 16 B).
    std::vector<RlF4> blob;
    RlSceneLayout lay;
    std::memset(&lay, 0, sizeof lay);
    auto append = [&](const std::vector<RlF4>& v) {
        const uint32_t off = (uint32_t)blob.size();
        blob.insert(blob.end(), v.begin(), v.end());
        return off;
    };
    append(fs.spheres);
    for (size_t pos = fs.cluster_base; pos < fs.spheres.size(); ++pos) blob[pos].w = fs.sphere_cull_w[pos]; // see RlSceneView::sphere_r2
    // the tables (RlSceneLayout: records [off_planes, off_objects)) ...
    lay.off_planes = append(fs.planes);
    lay.off_parabs = append(fs.parabs);
    lay.off_prisms = append(fs.prisms);
    lay.off_cull = append(fs.cull_bounds);
    lay.off_prism_cyl = append(fs.prism_cyl);
    lay.prism_cylinders = fs.prism_cylinders ? 1u : 0u;
    lay.group_gc = fs.group_gc;
    lay.small_ordered = (fs.small_ordered ? 1u : 0u) | (fs.small_axis_z ? 2u : 0u); // (bit 1: rl_scan_wave, the six small primitives of the built-in room)
    lay.off_camera = append(fs.camera_rec);
    lay.cull_cmax2 = fs.cull_cmax2;
    // ... then the per-object arrays
    lay.off_objects = append(fs.objects);
    // (an object's group bits hold 26 bits -- a blob index on the device -- and rl_object_bits masks silently)
    if (blob.size() >= (1u << 26) || fs.cluster_base + fs.spheres.size() >= (1u << 26))
        return fail(RL_E_INVALID, "scene too large: its records do not fit the 26 index bits of an object record");
    if (fs.n_clusters != 0 && fs.cluster_k != 10u && fs.cluster_k != 14u) {
#ifndef RL_CLUSTER_K
        return fail(RL_E_INVALID, "cluster size is not one the trace kernel's member rounds are unrolled for (RL_CLUSTER_K_CHOICES)");
#endif
    }
    // (on the device an object's group bits are the blob index of the record its hit is completed from, RlSceneView::records)
    for (size_t i = 0; i < fs.objects.size(); ++i) {
        const uint32_t bits = rl_f2u(fs.objects[i].w), kind = rl_object_surface(bits), group = rl_object_group(bits);
        const uint32_t index = kind == RL_SURFACE_SPHERE       ? group
                               : kind == RL_SURFACE_PARABOLOID ? lay.off_parabs + 3u * group
                               : kind == RL_SURFACE_HEX_PRISM  ? lay.off_prisms + (uint32_t)RL_PRISM_STRIDE * group
                                                               : lay.off_planes + 2u * group;
        blob[lay.off_objects + i].w = rl_u2f(rl_object_bits(kind, rl_object_material(bits), index));
    }
    lay.off_cie = (uint32_t)blob.size();
    const RlF4* cie = (const RlF4*)RL_CIE1931_XYZ0;
    blob.insert(blob.end(), cie, cie + RL_CIE_SAMPLES);
    lay.off_sphere_obj = (uint32_t)blob.size();
    const size_t so_f4 = (fs.sphere_obj.size() + 3) / 4;
    blob.resize(blob.size() + so_f4);
    std::memcpy(blob.data() + lay.off_sphere_obj, fs.sphere_obj.data(), fs.sphere_obj.size() * sizeof(uint32_t));
    lay.off_sphere_r2 = (uint32_t)blob.size();
    blob.resize(blob.size() + so_f4);
    for (size_t pos = 0; pos < fs.spheres.size(); ++pos) ((float*)(blob.data() + lay.off_sphere_r2))[pos] = fs.spheres[pos].w;
    lay.total_f4 = (uint32_t)blob.size();
    lay.n_direct = fs.n_direct;
    lay.n_direct_padded = fs.n_direct_padded;
    lay.cluster_base = fs.cluster_base;
    lay.n_clusters = fs.n_clusters;
    lay.cluster_k = fs.cluster_k;
    lay.n_cluster_groups = fs.n_cluster_groups;
    lay.n_prism_groups = fs.n_prism_groups;
    lay.n_cluster_supers = fs.n_cluster_supers;
    lay.super_g = fs.super_g;
    lay.n_planes = (uint32_t)(fs.planes.size() / 2);
    lay.n_parabs = (uint32_t)(fs.parabs.size() / 3);
    lay.n_prisms = (uint32_t)(fs.prisms.size() / RL_PRISM_STRIDE);
    lay.n_objects = (uint32_t)fs.objects.size();

    RlScene* s = new (std::nothrow) RlScene();
    if (!s) return fail(RL_E_INVALID, "out of host memory");
    s->device = device;
    s->lay = lay;
    s->staged_bytes = blob.size() * sizeof(RlF4);
    s->tables_bytes = (size_t)(lay.off_objects - lay.off_planes) * sizeof(RlF4);
    s->blob = nullptr;
    hipError_t e = hipMalloc((void**)&s->blob, s->staged_bytes);
    if (e == hipSuccess) e = hipMemcpy(s->blob, blob.data(), s->staged_bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (s->blob) (void)hipFree(s->blob);
        delete s;
        return fail(RL_E_HIP, std::string("scene upload: ") + hipGetErrorString(e));
    }
    *out = s;
    return RL_OK;
}

namespace {
int sessions_quiesce(int device);
int render_end(RlTraceUnit* u);
int plot_settle(RlPlotUnit* plot);
}

int rl_scene_destroy(RlScene* scene) {
    if (!scene) return RL_OK;
    (void)hipSetDevice(scene->device);
    (void)sessions_quiesce(scene->device); // an open launch may still be reading the blob
    (void)hipFree(scene->blob);
    delete scene;
    return RL_OK;
}

// ---- TraceUnit ----------------------------------------------------------------------------------

int rl_trace_unit_create(int device, uint32_t id, uint32_t width, uint32_t height, uint32_t n_photons,
                         RlTraceUnit** out) {
    if (!out) return fail(RL_E_INVALID, "null output handle");
    *out = nullptr;
    if (width == 0 || height == 0 || n_photons == 0) return fail(RL_E_INVALID, "zero-sized trace unit");
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    RlTraceUnit* u = new (std::nothrow) RlTraceUnit();
    if (!u) return fail(RL_E_INVALID, "out of host memory");
    u->device = device;
    u->id = id;
    u->width = width;
    u->height = height;
    u->n_photons = n_photons;
    u->photons = nullptr;
    u->queue = nullptr;
    u->stream = nullptr;
    u->rendered = nullptr;
    u->fetch = RL_FETCH_LDS;
    u->kernel_ms = 0.0;
    u->launches = 0;
    u->tuned_dyn = 0;
    u->tuned_kernel = nullptr;
    u->tuned_per_cu = 0;
    u->cu_count = 256;
    hipError_t e = hipMalloc((void**)&u->photons, (size_t)n_photons * sizeof(RlMappedPhoton));
    if (e == hipSuccess) e = hipMemset(u->photons, 0, (size_t)n_photons * sizeof(RlMappedPhoton)); // MappedPhoton::new
    if (e == hipSuccess) e = hipMalloc((void**)&u->queue, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(u->queue, 0, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&u->stream, getenv("RL_BLOCKING_STREAMS") ? hipStreamDefault : hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&u->rendered, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // the memsets above ran on the null stream
    if (e != hipSuccess) {
        rl_trace_unit_destroy(u);
        return fail(RL_E_HIP, std::string("trace unit allocation: ") + hipGetErrorString(e));
    }
    if ((rc = cu_count_of(device, &u->cu_count)) != RL_OK) {
        rl_trace_unit_destroy(u);
        return rc;
    }
    *out = u;
    return RL_OK;
}

int rl_trace_unit_destroy(RlTraceUnit* u) {
    if (!u) return RL_OK;
    (void)hipSetDevice(u->device);
    (void)render_end(u); // a render that was begun writes to this unit's photons until it is complete
    if (u->stream) (void)hipStreamSynchronize(u->stream);
    for (EventPair& ep : u->pending) {
        (void)hipEventDestroy(ep.start);
        (void)hipEventDestroy(ep.stop);
    }
    for (EventPair& ep : u->pool) {
        (void)hipEventDestroy(ep.start);
        (void)hipEventDestroy(ep.stop);
    }
    if (u->stream) (void)hipStreamDestroy(u->stream);
    if (u->rendered) (void)hipEventDestroy(u->rendered);
    if (u->photons) (void)hipFree(u->photons);
    if (u->queue) (void)hipFree(u->queue);
    delete u;
    return RL_OK;
}

int rl_trace_unit_set_fetch(RlTraceUnit* u, int primitive_fetch) {
    if (!u) return fail(RL_E_INVALID, "null trace unit");
    if (primitive_fetch != RL_FETCH_LDS && primitive_fetch != RL_FETCH_GLOBAL) return fail(RL_E_INVALID, "unknown fetch mode");
    u->fetch = primitive_fetch;
    return RL_OK;
}

int rl_trace_unit_render_async(RlTraceUnit* u, const RlScene* scene, uint64_t seed, uint32_t stream, uint64_t first_path_index) {
    if (!u || !scene) return fail(RL_E_INVALID, "null handle");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    return launch_trace(u, scene, u->photons, nullptr, seed, stream, first_path_index, u->n_photons);
}

namespace {

// ---- sessions: open launches (rl_kernels.hip.h, RlOpenDev / RlOpenCtl) ---------------------------------------------
// Blocking render calls of at most RL_SESSION_MAX_PATHS paths are APPENDED to a trace kernel that is already running
// on the device whenever there is one that still accepts work (same scene, seed, stream, image size, fetch mode,
// fused or not); otherwise the call starts such a kernel with itself as the first job.  Each call returns as soon as
// ITS paths are finished -- the kernel goes on with the other callers' -- so the drain tail of one 524,288-path batch
// overlaps the next batches instead of idling the chip, and nothing is launched per call.  A kernel stays open while
// calls are still finishing their last paths (their callers come back with more) and for a grace period after the
// last one, then closes itself; a call that arrives too late for it starts the next one, whose workgroups move onto
// the CUs as the closed one's drain.  The calls also exist in two halves (begin / end, `Ticket`): whoever uses a
// begun render's result next ends it.

struct Session {
    hipStream_t stream = nullptr;
    RlOpenDev* od = nullptr;
    RlOpenCtl* ctl = nullptr;     // pinned, coherent host memory
    RlOpenCtl* ctl_dev = nullptr; // the same memory as the device addresses it
    unsigned long long* counters = nullptr; // [unused, segments, paths] of the running kernel
    hipEvent_t start = nullptr, stop = nullptr;
    bool launched = false; // a kernel was launched and its counters are not harvested yet
    bool open = false;     // the host may still try to append
    uint32_t n = 0;        // jobs appended so far
    int waiters = 0;       // calls that have not yet read their completion flag (ctl must not be recycled under them)
    const void* tuned_kernel = nullptr; // launch configuration last set up for this slot
    size_t tuned_dyn = 0;
    int tuned_per_cu = 1;
    const RlScene* scene = nullptr;
    uint64_t seed = 0;
    uint32_t stream_id = 0, width = 0, height = 0;
    int fetch = 0;
    bool fused = false;
    // who the running kernel works for: (a unit's counters, paths of its call) per job; its run time is split by paths
    std::vector<std::pair<std::shared_ptr<UnitCounters>, uint64_t>> shares;
};

struct DeviceSessions {
    std::mutex lock;
    std::condition_variable changed;
    Session s[4];
    bool ready = false;
    uint64_t launches = 0;
    uint64_t histogram[RL_OPEN_CAP + 1] = {}; // finished kernels by number of calls they carried
    double presync_us = 0.0, admit_us = 0.0, wait_us = 0.0; // where the calls spent their time (RL_OPEN_LAUNCH_TIMING=1 prints it)
    uint64_t calls = 0, starts = 0;
};

DeviceSessions* sessions_of(int device) {
    static DeviceSessions all[64];
    return &all[device >= 0 && device < 64 ? device : 0];
}

inline uint32_t host_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }

int sessions_setup(DeviceSessions* d) { // under d->lock, device current
    if (d->ready) return RL_OK;
    int least = 0, greatest = 0;
    RL_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (Session& x : d->s) {
        // a priority of its own = a hardware queue of its own: the streams of the plot / gather / tonemap kernels must
        // never queue up behind a resident trace kernel (they run beside it in the registers it leaves free)
        if (hipStreamCreateWithPriority(&x.stream, hipStreamNonBlocking, least) != hipSuccess) { // (no priorities here: a plain stream)
            (void)hipGetLastError();
            RL_HIP(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
        }
        RL_HIP(hipMalloc((void**)&x.od, sizeof(RlOpenDev)));
        RL_HIP(hipHostMalloc((void**)&x.ctl, sizeof(RlOpenCtl), hipHostMallocCoherent | hipHostMallocMapped));
        RL_HIP(hipHostGetDevicePointer((void**)&x.ctl_dev, x.ctl, 0));
        RL_HIP(hipMalloc((void**)&x.counters, 3 * sizeof(unsigned long long)));
        RL_HIP(hipEventCreate(&x.start));
        RL_HIP(hipEventCreate(&x.stop));
    }
    d->ready = true;
    return RL_OK;
}

// The kernel of session x has ended (or was never launched): add its counters to the device's tally.
int session_harvest(DeviceSessions* d, Session& x) {
    if (!x.launched) return RL_OK;
    RL_HIP(hipStreamSynchronize(x.stream));
    float ms = 0.0f;
    RL_HIP(hipEventElapsedTime(&ms, x.start, x.stop));
    uint64_t total = 0;
    for (const auto& sh : x.shares) total += sh.second;
    for (const auto& sh : x.shares) // every unit gets the part of the kernel's time that its paths are of the kernel's paths
        if (total) sh.first->kernel_ns += (uint64_t)((double)ms * 1.0e6 * ((double)sh.second / (double)total));
    x.shares.clear();
    d->launches += 1;
    const uint32_t carried = host_load(&x.ctl->final_at);
    d->histogram[carried <= RL_OPEN_CAP ? carried : 0] += 1;
    x.launched = false;
    x.open = false;
    return RL_OK;
}

// Waits until no session kernel runs on `device` and credits their run time to the units they worked for.  The waiting
// happens OUTSIDE the device's lock: other threads' render calls go on being admitted meanwhile -- to the
// kernels that are running, or to new ones, which this call then does not wait for.
int sessions_quiesce(int device) {
    DeviceSessions* d = sessions_of(device);
    hipStream_t streams[4];
    int n = 0;
    {
        std::lock_guard<std::mutex> guard(d->lock);
        if (!d->ready) return RL_OK;
        for (Session& x : d->s)
            if (x.launched) streams[n++] = x.stream;
    }
    for (int i = 0; i < n; ++i) RL_HIP(hipStreamSynchronize(streams[i]));
    std::lock_guard<std::mutex> guard(d->lock);
    for (Session& x : d->s) {
        if (!x.launched || hipStreamQuery(x.stream) != hipSuccess) continue; // restarted meanwhile: its next harvest counts it
        const int rc = session_harvest(d, x);
        if (rc != RL_OK) return rc;
    }
    return RL_OK;
}

int session_start(DeviceSessions* d, Session& x, RlTraceUnit* u, const RlScene* scene, bool fused, uint64_t seed, uint32_t stream_id,
                  const RlJobEntry& first) {
    int rc = session_harvest(d, x);
    if (rc != RL_OK) return rc;
    x.scene = scene;
    x.seed = seed;
    x.stream_id = stream_id;
    x.width = u->width;
    x.height = u->height;
    x.fetch = u->fetch;
    x.fused = fused;
    RlTraceJob job;
    job.width = u->width;
    job.height = u->height;
    job.aspect_ratio = (float)u->width / (float)u->height;
    job.stream = stream_id;
    job.seed = seed;
    job.first_path = 0;
    job.n_paths = 0;
    static const uint32_t grace_us = getenv("RL_OPEN_LAUNCH_GRACE_US") ? (uint32_t)atoi(getenv("RL_OPEN_LAUNCH_GRACE_US")) : 150u;
    job.grace_ticks = grace_us * 100u; // ticks of wall_clock64() (100 MHz) an idle open launch waits for another call
    job.reserved = 0;
    job.wm1 = (float)(int)u->width - 1.0f;
    job.hm1 = (float)(int)u->height - 1.0f;
    const size_t scratch_bytes = (RL_TRACE_BLOCK / 64) * sizeof(RlWaveScratch) + sizeof(RlOpenWg) + ring_t_bytes(scene);
    size_t blob_bytes = 0;
    const int stage = stage_of(scene, u->fetch, scratch_bytes, &blob_bytes);
    auto kernel = trace_kernel_variant(stage, fused, true, scene->lay.prism_cylinders != 0u);
    const size_t dyn = scratch_bytes + blob_bytes;
    if (x.tuned_kernel != (const void*)kernel || x.tuned_dyn != dyn) { // once per (slot, variant, scene size)
        RL_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int n = 1;
        RL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, RL_TRACE_BLOCK, dyn));
        x.tuned_per_cu = n < 1 ? 1 : n;
        x.tuned_kernel = (const void*)kernel;
        x.tuned_dyn = dyn;
    }
    const int per_cu = x.tuned_per_cu;
    std::memset(x.ctl, 0, sizeof(RlOpenCtl));
    x.ctl->jobs[0] = first;
    x.ctl->closed_at = RL_OPEN_NONE;
    x.ctl->final_at = RL_OPEN_NONE;
    __atomic_store_n(&x.ctl->published, 1u, __ATOMIC_SEQ_CST);
    x.n = 1;
    RL_HIP(hipMemsetAsync(x.od, 0, sizeof(RlOpenDev), x.stream));
    RL_HIP(hipMemsetAsync(x.counters, 0, 3 * sizeof(unsigned long long), x.stream));
    RL_HIP(hipEventRecord(x.start, x.stream));
    hipLaunchKernelGGL(kernel, dim3((unsigned)(u->cu_count * per_cu)), dim3(RL_TRACE_BLOCK), dyn, x.stream, scene->blob, scene->lay, job,
                       (RlMappedPhoton*)nullptr, (float*)nullptr, x.counters, (const RlJobEntry*)x.od->jobs, x.od, x.ctl_dev);
    RL_HIP(hipGetLastError());
    RL_HIP(hipEventRecord(x.stop, x.stream));
    x.launched = true;
    x.open = true;
    return RL_OK;
}

// Appends `e` to the running kernel of x.  Returns the job index, or -1 when the kernel had already closed.
int session_append(Session& x, const RlJobEntry& e) {
    const uint32_t k = x.n;
    x.ctl->jobs[k] = e;
    __atomic_store_n(&x.ctl->done[k], 0u, __ATOMIC_RELAXED);
    __atomic_store_n(&x.ctl->published, k + 1u, __ATOMIC_SEQ_CST);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    for (uint64_t spins = 0;; ++spins) {
        const uint32_t closed_at = host_load(&x.ctl->closed_at);
        if (closed_at == RL_OPEN_NONE || closed_at > k) break; // it had not closed when the entry became visible, or re-opened
        const uint32_t final_at = host_load(&x.ctl->final_at);
        if (final_at != RL_OPEN_NONE && final_at <= k) return -1;
        if (spins > 1000000 && hipStreamQuery(x.stream) != hipErrorNotReady) return -1; // the kernel is gone (an error surfaces later)
        __builtin_ia32_pause();
    }
    x.n = k + 1u;
    return (int)k;
}

// Blocks until job k of session x is complete.
int session_wait(Session& x, uint32_t k) {
    auto last_query = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
        if (host_load(&x.ctl->done[k]) != 0u) return RL_OK;
        if (spins < 4000) {
            __builtin_ia32_pause();
        } else if ((spins & 255u) == 0u && std::chrono::steady_clock::now() - last_query > std::chrono::milliseconds(20)) {
            // rarely (the query is not cheap while a kernel runs): has the kernel died or ended under the call?
            last_query = std::chrono::steady_clock::now();
            const hipError_t q = hipStreamQuery(x.stream);
            if (q == hipSuccess) { // the kernel has ended: it reported every accepted job before that
                if (host_load(&x.ctl->done[k]) != 0u) return RL_OK;
                return fail(RL_E_STATE, "an open trace launch ended without completing one of its calls");
            }
            if (q != hipErrorNotReady) return fail(RL_E_HIP, std::string("open trace launch: ") + hipGetErrorString(q));
        } else if (spins < 4000 + 64) {
            std::this_thread::yield();
        } else {
            // Round 6: a call that is not done after ~50 us of spinning and yielding SLEEPS between looks.  Yielding for ever is fine
            // while there are at most as many waiting workers as cores; the reference starts num_cpus::get() workers (app.rs:55), a
            // host may have fewer cores than that per GPU, and waiting threads that keep yielding then take the cores from the ones
            // that have tasks to issue (un-fused, workers that wait for their own batch: 14.1 Grays/s at 16 workers, 5.3 at 64 on 16 cores).
            static thread_local bool slack_set = false;
            if (!slack_set) {
                prctl(PR_SET_TIMERSLACK, 2000UL, 0UL, 0UL, 0UL); // (the default slack of 50 us would triple a 25 us sleep)
                slack_set = true;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(25));
        }
    }
}

// First half of a blocking render served by an open launch: appends the call (or starts a launch with it) and leaves a
// ticket in the unit.
int session_begin(RlTraceUnit* u, const RlScene* scene, RlPlotUnit* plot, uint64_t seed, uint32_t stream_id, uint64_t first_path_index,
                  uint64_t n_paths) {
    // What the call's target is still being read or cleared by (a plot of the unit's previous photons, the gather's
    // clear of the plot buffer) must be over before a kernel that is already running may write to it.
    const auto t0 = std::chrono::steady_clock::now();
    if (plot) {
        RL_HIP(hipEventSynchronize(plot->cleared));
        // ... and whatever else still reads or writes the buffer on the plot unit's own stream (rl_plot_unit_add, the RCCL
        // reduce, an upload): a splat of the running kernel must not race them
        RL_HIP(hipStreamSynchronize(plot->stream));
    } else {
        RL_HIP(hipStreamSynchronize(u->stream));
    }
    const auto t1 = std::chrono::steady_clock::now();
    RlJobEntry e;
    e.target = plot ? (void*)plot->xyz : (void*)u->photons;
    e.first_path = first_path_index;
    e.start = 0;
    e.end = n_paths;
    DeviceSessions* d = sessions_of(u->device);
    Session* mine = nullptr;
    int k = -1;
    {
        std::unique_lock<std::mutex> guard(d->lock);
        int rc = sessions_setup(d);
        if (rc != RL_OK) return rc;
        for (;;) {
            for (Session& x : d->s) // a running kernel that takes this call?
                if (x.open && x.n < RL_OPEN_CAP && x.scene == scene && x.seed == seed && x.stream_id == stream_id && x.width == u->width &&
                    x.height == u->height && x.fetch == u->fetch && x.fused == (plot != nullptr)) {
                    k = session_append(x, e);
                    if (k >= 0) mine = &x, x.waiters += 1, x.shares.emplace_back(u->counters, n_paths);
                    else x.open = false;
                    break;
                }
            if (mine) break;
            for (Session& x : d->s) // a kernel closes itself; the host learns of it here (or when an append is turned down)
                if (x.open && host_load(&x.ctl->final_at) != RL_OPEN_NONE) x.open = false;
            for (Session& x : d->s) // no: start one in a slot whose kernel has ended
                if (!x.open && x.waiters == 0 && (!x.launched || hipStreamQuery(x.stream) == hipSuccess)) {
                    rc = session_start(d, x, u, scene, plot != nullptr, seed, stream_id, e);
                    if (rc != RL_OK) return rc;
                    mine = &x;
                    x.waiters += 1;
                    x.shares.emplace_back(u->counters, n_paths);
                    k = 0;
                    break;
                }
            if (mine) break;
            bool retry = false; // an open session that is full: close it to the host
            for (Session& x : d->s)
                if (x.open && x.n >= RL_OPEN_CAP) x.open = false, retry = true;
            if (retry) continue;
            // Every slot is taken by another (scene, seed, stream, size, fetch, fused) combination.  A slot nobody waits
            // on frees itself (its kernel closes after a grace period), but one that carries begun renders stays until
            // they are ended -- possibly by this very thread (a fifth combination begun without ending an
            // earlier one used to spin here for ever).  If all of them do, the call gets a launch of its own instead.
            bool all_pinned = true;
            for (Session& x : d->s)
                if (x.waiters == 0) all_pinned = false;
            if (all_pinned) return RL_SESSIONS_PINNED;
            d->changed.wait_for(guard, std::chrono::microseconds(50));
        }
    }
    Ticket& t = plot ? plot->ticket : u->ticket;
    t.pending = true;
    t.session = mine;
    t.job = (uint32_t)k;
    t.paths = n_paths;
    t.device = u->device;
    t.counters = u->counters;
    t.presync_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    t.admit_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    return RL_OK;
}

// Second half: waits until the call's paths are finished.
int session_end(Ticket& t) {
    Session* mine = t.session;
    const uint32_t k = t.job;
    DeviceSessions* d = sessions_of(t.device);
    const auto t2 = std::chrono::steady_clock::now();
    int rc = session_wait(*mine, k);
    const uint64_t job_segments = host_load(&mine->ctl->segs[k]);
    {
        const auto t3 = std::chrono::steady_clock::now();
        std::lock_guard<std::mutex> guard(d->lock);
        mine->waiters -= 1;
        d->changed.notify_all(); // a slot may have become free for a thread waiting in session_begin
        d->presync_us += t.presync_us;
        d->admit_us += t.admit_us;
        d->wait_us += std::chrono::duration<double, std::micro>(t3 - t2).count();
        d->calls += 1;
    }
    t.session = nullptr;
    t.pending = false;
    if (rc == RL_OK) {
        t.counters->paths += t.paths;
        t.counters->segments += job_segments;
    }
    t.counters.reset();
    return rc;
}

#define RL_SESSION_MAX_PATHS (1ull << 28) // per call: its segment count must fit 32 bits

// The two halves of the blocking render of both kinds: un-fused (plot == nullptr, n_paths = the unit's batch) and fused.
int render_begin(RlTraceUnit* u, const RlScene* scene, RlPlotUnit* plot, uint64_t seed, uint32_t stream, uint64_t first_path_index,
                 uint64_t n_paths) {
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if (!plot && u->ticket.pending) return fail(RL_E_STATE, "the trace unit has a render that was begun and not ended");
    if (plot && (rc = plot_settle(plot)) != RL_OK) return rc; // the plot unit's previous begun render first
    if (n_paths == 0) return RL_OK;
    if (scene->device != u->device) return fail(RL_E_STATE, "scene and trace unit live on different devices");
    if (first_path_index + n_paths < first_path_index || first_path_index + n_paths == ~0ull)
        return fail(RL_E_INVALID, "path indices must stay below 2^64 - 1");
    if (n_paths % 64 == 0 && n_paths < RL_SESSION_MAX_PATHS) {
        rc = session_begin(u, scene, plot, seed, stream, first_path_index, n_paths);
        if (rc != RL_SESSIONS_PINNED) return rc;
    }
    // a ragged or a huge batch, or no open launch to be had (rl_trace_unit_render_begin in robigo_luculenta.h): a launch of
    // its own on the unit's stream (a fused one is waited for by the plot unit's stream)
    rc = launch_trace(u, scene, plot ? nullptr : u->photons, plot, seed, stream, first_path_index, n_paths);
    if (rc != RL_OK) return rc;
    Ticket& t = plot ? plot->ticket : u->ticket;
    t.pending = true;
    t.session = nullptr;
    t.device = u->device;
    return RL_OK;
}

// Ends the fused render that splats into `plot`, if one was begun: everything that reads, clears or adds to the buffer
// calls this first.
int plot_settle(RlPlotUnit* plot) {
    Ticket& t = plot->ticket;
    if (!t.pending) return RL_OK;
    int rc = use_device(plot->device);
    if (rc != RL_OK) return rc;
    if (t.session) return session_end(t);
    t.pending = false;
    RL_HIP(hipStreamSynchronize(plot->stream)); // it waits for the launch (launch_trace)
    return RL_OK;
}

// Ends the un-fused render begun on `u`, if any.
int render_end(RlTraceUnit* u) {
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    Ticket& t = u->ticket;
    if (!t.pending) return RL_OK;
    if (t.session) {
        if ((rc = session_end(t)) != RL_OK) return rc;
        RL_HIP(hipEventRecord(u->rendered, u->stream)); // PlotUnit::plot waits for this (complete already)
        return RL_OK;
    }
    t.pending = false;
    RL_HIP(hipStreamSynchronize(u->stream));
    return RL_OK;
}

} // namespace

// TraceUnit::render as the reference's workers call it: blocking, one 524,288-path batch per call (trace_unit.rs:67).
// That batch is 0.15 ms of MI355X work followed by ~0.2 ms in which the launch waits for its longest paths, so the
// calls that are in flight at the same time -- the reference runs one per worker thread -- share OPEN launches (above).
// Results are those of separate launches, bit for bit: a path is a pure function of (seed, stream, path index).
int rl_trace_unit_render(RlTraceUnit* u, const RlScene* scene, uint64_t seed, uint32_t stream, uint64_t first_path_index) {
    if (!u || !scene) return fail(RL_E_INVALID, "null handle");
    const int rc = render_begin(u, scene, nullptr, seed, stream, first_path_index, u->n_photons);
    return rc != RL_OK ? rc : render_end(u);
}

int rl_trace_unit_render_begin(RlTraceUnit* u, const RlScene* scene, uint64_t seed, uint32_t stream, uint64_t first_path_index) {
    if (!u || !scene) return fail(RL_E_INVALID, "null handle");
    return render_begin(u, scene, nullptr, seed, stream, first_path_index, u->n_photons);
}

int rl_trace_unit_render_end(RlTraceUnit* u) {
    if (!u) return fail(RL_E_INVALID, "null trace unit");
    return render_end(u);
}

namespace {
int check_fused(const RlTraceUnit* u, const RlScene* scene, const RlPlotUnit* plot) {
    if (!u || !scene || !plot) return fail(RL_E_INVALID, "null handle");
    if (plot->device != u->device || plot->width != u->width || plot->height != u->height)
        return fail(RL_E_STATE, "plot unit does not match the trace unit (device or size)");
    return RL_OK;
}
} // namespace

int rl_trace_unit_render_fused_sync(RlTraceUnit* u, const RlScene* scene, RlPlotUnit* plot, uint64_t seed, uint32_t stream,
                                    uint64_t first_path_index, uint64_t n_paths) {
    int rc = check_fused(u, scene, plot);
    if (rc == RL_OK) rc = render_begin(u, scene, plot, seed, stream, first_path_index, n_paths);
    return rc != RL_OK ? rc : plot_settle(plot);
}

int rl_trace_unit_render_fused_begin(RlTraceUnit* u, const RlScene* scene, RlPlotUnit* plot, uint64_t seed, uint32_t stream,
                                     uint64_t first_path_index, uint64_t n_paths) {
    const int rc = check_fused(u, scene, plot);
    return rc != RL_OK ? rc : render_begin(u, scene, plot, seed, stream, first_path_index, n_paths);
}

int rl_trace_unit_render_fused(RlTraceUnit* u, const RlScene* scene, RlPlotUnit* plot, uint64_t seed, uint32_t stream,
                               uint64_t first_path_index, uint64_t n_paths) {
    if (!u || !scene || !plot) return fail(RL_E_INVALID, "null handle");
    if (plot->device != u->device || plot->width != u->width || plot->height != u->height)
        return fail(RL_E_STATE, "plot unit does not match the trace unit (device or size)");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    return launch_trace(u, scene, nullptr, plot, seed, stream, first_path_index, n_paths);
}

int rl_trace_unit_sync(RlTraceUnit* u) {
    if (!u) return fail(RL_E_INVALID, "null trace unit");
    int rc = render_end(u); // also ends a render that was begun
    if (rc != RL_OK) return rc;
    RL_HIP(hipStreamSynchronize(u->stream));
    return drain_events(u);
}

int rl_trace_unit_photons(RlTraceUnit* u, RlMappedPhoton* out) {
    if (!u || !out) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = render_end(u)) != RL_OK) return rc; // a render that was begun: its photons are complete after this
    RL_HIP(hipStreamSynchronize(u->stream));
    RL_HIP(hipMemcpy(out, u->photons, (size_t)u->n_photons * sizeof(RlMappedPhoton), hipMemcpyDeviceToHost));
    return RL_OK;
}

int rl_trace_unit_stats(RlTraceUnit* u, uint64_t* paths, uint64_t* segments, double* kernel_ms) {
    if (!u) return fail(RL_E_INVALID, "null trace unit");
    int rc = render_end(u); // renders that were begun are counted when they end
    if (rc != RL_OK) return rc;
    RL_HIP(hipStreamSynchronize(u->stream));
    if ((rc = drain_events(u)) != RL_OK) return rc;
    unsigned long long q[3];
    RL_HIP(hipMemcpy(q, u->queue, sizeof q, hipMemcpyDeviceToHost));
    // Calls served by open launches are counted per call (session_*); the run time of such a launch, which serves many
    // units at once, is split among them by paths when it has ended (session_harvest).
    if ((rc = sessions_quiesce(u->device)) != RL_OK) return rc;
    if (segments) *segments = q[1] + u->counters->segments.load();
    if (paths) *paths = q[2] + u->counters->paths.load();
    if (kernel_ms) *kernel_ms = u->kernel_ms + (double)u->counters->kernel_ns.load() * 1.0e-6;
    return RL_OK;
}

// ---- PlotUnit -----------------------------------------------------------------------------------

int rl_plot_unit_create(int device, uint32_t id, uint32_t width, uint32_t height, float* external_xyz, RlPlotUnit** out) {
    if (!out) return fail(RL_E_INVALID, "null output handle");
    *out = nullptr;
    if (width == 0 || height == 0) return fail(RL_E_INVALID, "zero-sized plot unit");
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    RlPlotUnit* u = new (std::nothrow) RlPlotUnit();
    if (!u) return fail(RL_E_INVALID, "out of host memory");
    u->device = device;
    u->id = id;
    u->width = width;
    u->height = height;
    u->xyz = external_xyz;
    u->owns = external_xyz == nullptr;
    u->cie = nullptr;
    u->stream = nullptr;
    u->plotted = u->ready = u->cleared = u->tail = nullptr;
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    hipError_t e = hipSuccess;
    if (u->owns) {
        e = hipMalloc((void**)&u->xyz, bytes);
        if (e == hipSuccess) e = hipMemset(u->xyz, 0, bytes);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&u->cie, sizeof RL_CIE1931_XYZ0);
    if (e == hipSuccess) e = hipMemcpy(u->cie, RL_CIE1931_XYZ0, sizeof RL_CIE1931_XYZ0, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&u->stream, getenv("RL_BLOCKING_STREAMS") ? hipStreamDefault : hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&u->plotted, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&u->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&u->cleared, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&u->tail, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // the synchronous memsets / copies above ran on the null stream
    if (e != hipSuccess) {
        rl_plot_unit_destroy(u);
        return fail(RL_E_HIP, std::string("plot unit allocation: ") + hipGetErrorString(e));
    }
    *out = u;
    return RL_OK;
}

int rl_plot_unit_destroy(RlPlotUnit* u) {
    if (!u) return RL_OK;
    (void)hipSetDevice(u->device);
    (void)plot_settle(u); // a fused render that was begun splats into this buffer until it is complete
    if (u->stream) {
        (void)hipStreamSynchronize(u->stream);
        (void)hipStreamDestroy(u->stream);
    }
    if (u->plotted) (void)hipEventDestroy(u->plotted);
    if (u->ready) (void)hipEventDestroy(u->ready);
    if (u->cleared) (void)hipEventDestroy(u->cleared);
    if (u->tail) (void)hipEventDestroy(u->tail);
    for (EventPair& ep : u->exchanges) (void)hipEventDestroy(ep.start), (void)hipEventDestroy(ep.stop);
    for (EventPair& ep : u->exchange_pool) (void)hipEventDestroy(ep.start), (void)hipEventDestroy(ep.stop);
    if (u->owns && u->xyz) (void)hipFree(u->xyz);
    if (u->cie) (void)hipFree(u->cie);
    delete u;
    return RL_OK;
}

int rl_plot_unit_plot(RlPlotUnit* u, RlTraceUnit* const* trace_units, uint32_t n_trace_units) {
    if (!u || (!trace_units && n_trace_units)) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(u)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    int cus = 256;
    if ((rc = cu_count_of(u->device, &cus)) != RL_OK) return rc;
    const float aspect = (float)u->width / (float)u->height; // plot_unit.rs:48
    for (uint32_t k = 0; k < n_trace_units; ++k) {           // app.rs:138-140
        RlTraceUnit* t = trace_units[k];
        if (!t || t->device != u->device) return fail(RL_E_STATE, "trace unit missing or on another device");
        if ((rc = render_end(t)) != RL_OK) return rc;          // a render that was begun: its photons are complete after this
        RL_HIP(hipStreamWaitEvent(u->stream, t->rendered, 0)); // mapped_photons complete (a no-op after a synchronous render)
        hipLaunchKernelGGL(rl_plot_kernel, dim3(grid_for(t->n_photons, cus)), dim3(RL_BLOCK), 0, u->stream, t->photons,
                           t->n_photons, u->cie, u->width, u->height, aspect, u->xyz);
        RL_HIP(hipGetLastError());
    }
    // The trace units may be handed out again as soon as this returns (task_scheduler.rs:262-271): their
    // next render must not overwrite mapped_photons before the kernels above have read them.
    if (n_trace_units != 0) {
        RL_HIP(hipEventRecord(u->plotted, u->stream));
        for (uint32_t k = 0; k < n_trace_units; ++k) RL_HIP(hipStreamWaitEvent(trace_units[k]->stream, u->plotted, 0));
    }
    return RL_OK;
}

int rl_plot_unit_clear(RlPlotUnit* u) {
    if (!u) return fail(RL_E_INVALID, "null plot unit");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(u)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    RL_HIP(hipMemsetAsync(u->xyz, 0, (size_t)u->width * u->height * 3 * sizeof(float), u->stream));
    RL_HIP(hipEventRecord(u->cleared, u->stream)); // later fused renders wait for this
    return RL_OK;
}

int rl_plot_unit_sync(RlPlotUnit* u) {
    if (!u) return fail(RL_E_INVALID, "null plot unit");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(u)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    RL_HIP(hipStreamSynchronize(u->stream));
    return RL_OK;
}

int rl_plot_unit_device_buffer(RlPlotUnit* u, float** device_xyz) {
    if (!u || !device_xyz) return fail(RL_E_INVALID, "null argument");
    const int rc = plot_settle(u);
    if (rc != RL_OK) return rc;
    *device_xyz = u->xyz;
    return RL_OK;
}

int rl_plot_unit_upload(RlPlotUnit* u, const RlVector3* in) {
    if (!u || !in) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(u)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    RL_HIP(hipStreamSynchronize(u->stream));
    RL_HIP(hipMemcpy(u->xyz, in, (size_t)u->width * u->height * sizeof(RlVector3), hipMemcpyHostToDevice));
    RL_HIP(hipStreamSynchronize(nullptr)); // (the null stream only: a resident open trace kernel of another unit is not waited for)
    return RL_OK;
}

int rl_plot_unit_download(RlPlotUnit* u, RlVector3* out) {
    if (!u || !out) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(u)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    RL_HIP(hipStreamSynchronize(u->stream));
    RL_HIP(hipMemcpy(out, u->xyz, (size_t)u->width * u->height * sizeof(RlVector3), hipMemcpyDeviceToHost));
    return RL_OK;
}

// ---- GatherUnit ---------------------------------------------------------------------------------

int rl_gather_unit_create(int device, uint32_t width, uint32_t height, RlGatherUnit** out) {
    if (!out) return fail(RL_E_INVALID, "null output handle");
    *out = nullptr;
    if (width == 0 || height == 0) return fail(RL_E_INVALID, "zero-sized gather unit");
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    RlGatherUnit* u = new (std::nothrow) RlGatherUnit();
    if (!u) return fail(RL_E_INVALID, "out of host memory");
    u->device = device;
    u->width = width;
    u->height = height;
    u->acc = u->comp = nullptr;
    u->stream = nullptr;
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    hipError_t e = hipMalloc((void**)&u->acc, bytes);
    if (e == hipSuccess) e = hipMemset(u->acc, 0, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&u->comp, bytes);
    if (e == hipSuccess) e = hipMemset(u->comp, 0, bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&u->stream, getenv("RL_BLOCKING_STREAMS") ? hipStreamDefault : hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // (the null stream only: a resident open trace kernel of another unit is not waited for)
    if (e != hipSuccess) {
        rl_gather_unit_destroy(u);
        return fail(RL_E_HIP, std::string("gather unit allocation: ") + hipGetErrorString(e));
    }
    *out = u;
    return RL_OK;
}

int rl_gather_unit_destroy(RlGatherUnit* u) {
    if (!u) return RL_OK;
    (void)hipSetDevice(u->device);
    if (u->stream) {
        (void)hipStreamSynchronize(u->stream);
        (void)hipStreamDestroy(u->stream);
    }
    if (u->acc) (void)hipFree(u->acc);
    if (u->comp) (void)hipFree(u->comp);
    delete u;
    return RL_OK;
}

int rl_gather_unit_accumulate(RlGatherUnit* u, RlPlotUnit* plot) {
    if (!u || !plot) return fail(RL_E_INVALID, "null handle");
    if (plot->device != u->device || plot->width != u->width || plot->height != u->height)
        return fail(RL_E_STATE, "plot unit does not match the gather unit (device or size)");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(plot)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    int cus = 256;
    if ((rc = cu_count_of(u->device, &cus)) != RL_OK) return rc;
    const uint64_t n_floats = (uint64_t)u->width * u->height * 3;
    RL_HIP(hipEventRecord(plot->ready, plot->stream));        // everything plotted / splatted / reduced into the buffer so far
    RL_HIP(hipStreamWaitEvent(u->stream, plot->ready, 0));
    hipLaunchKernelGGL(rl_gather_kernel, dim3(grid_for(n_floats / 4 + 1, cus)), dim3(RL_BLOCK), 0, u->stream, u->acc, u->comp,
                       plot->xyz, n_floats);
    RL_HIP(hipGetLastError());
    RL_HIP(hipEventRecord(plot->cleared, u->stream));         // the kernel also cleared the plot buffer (app.rs:147)
    RL_HIP(hipStreamWaitEvent(plot->stream, plot->cleared, 0));
    return RL_OK;
}

int rl_gather_unit_sync(RlGatherUnit* u) {
    if (!u) return fail(RL_E_INVALID, "null gather unit");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    RL_HIP(hipStreamSynchronize(u->stream));
    return RL_OK;
}

int rl_gather_unit_download(RlGatherUnit* u, RlVector3* tristimulus, RlVector3* compensation) {
    if (!u) return fail(RL_E_INVALID, "null gather unit");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    const size_t bytes = (size_t)u->width * u->height * sizeof(RlVector3);
    RL_HIP(hipStreamSynchronize(u->stream));
    if (tristimulus) RL_HIP(hipMemcpy(tristimulus, u->acc, bytes, hipMemcpyDeviceToHost));
    if (compensation) RL_HIP(hipMemcpy(compensation, u->comp, bytes, hipMemcpyDeviceToHost));
    return RL_OK;
}

int rl_gather_unit_save(RlGatherUnit* u, const char* path) {
    if (!u || !path) return fail(RL_E_INVALID, "null argument");
    const size_t n = (size_t)u->width * u->height;
    std::vector<RlVector3> host(2 * n);
    int rc = rl_gather_unit_download(u, host.data(), host.data() + n);
    if (rc != RL_OK) return rc;
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(RL_E_IO, std::string("failed to open file ") + path);
    const size_t written = std::fwrite(host.data(), sizeof(RlVector3), 2 * n, f);
    const bool flushed = std::fflush(f) == 0 && fsync(fileno(f)) == 0; // on disk before a caller renames it into place (rl_app.cpp)
    const int closed = std::fclose(f);
    if (written != 2 * n || closed != 0 || !flushed) return fail(RL_E_IO, std::string("failed to write raw buffer ") + path);
    return RL_OK;
}

int rl_gather_unit_load(RlGatherUnit* u, const char* path) {
    if (!u || !path) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(RL_E_IO, std::string("failed to open file ") + path);
    const size_t n = (size_t)u->width * u->height;
    std::vector<RlVector3> host(2 * n);
    // read_into semantics (read.rs:20-32): a short file leaves the tail as it was.
    if ((rc = rl_gather_unit_download(u, host.data(), host.data() + n)) != RL_OK) {
        std::fclose(f);
        return rc;
    }
    (void)std::fread(host.data(), 1, 2 * n * sizeof(RlVector3), f);
    std::fclose(f);
    RL_HIP(hipMemcpy(u->acc, host.data(), n * sizeof(RlVector3), hipMemcpyHostToDevice)); // the download above drained u->stream
    RL_HIP(hipMemcpy(u->comp, host.data() + n, n * sizeof(RlVector3), hipMemcpyHostToDevice));
    RL_HIP(hipStreamSynchronize(nullptr)); // (the null stream only: a resident open trace kernel of another unit is not waited for)
    return RL_OK;
}

// ---- TonemapUnit --------------------------------------------------------------------------------

int rl_tonemap_unit_create(int device, uint32_t width, uint32_t height, RlTonemapUnit** out) {
    if (!out) return fail(RL_E_INVALID, "null output handle");
    *out = nullptr;
    if (width == 0 || height == 0) return fail(RL_E_INVALID, "zero-sized tonemap unit");
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    RlTonemapUnit* u = new (std::nothrow) RlTonemapUnit();
    if (!u) return fail(RL_E_INVALID, "out of host memory");
    u->device = device;
    u->width = width;
    u->height = height;
    u->rgb = nullptr;
    u->srgb = nullptr;
    u->max_intensity = nullptr;
    u->last_stream = nullptr;
    const size_t n = (size_t)width * height * 3;
    hipError_t e = hipMalloc((void**)&u->rgb, n);
    if (e == hipSuccess) e = hipMemset(u->rgb, 0, n);
    if (e == hipSuccess) e = hipMalloc((void**)&u->srgb, n * sizeof(float));
    if (e == hipSuccess) e = hipMemset(u->srgb, 0, n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&u->max_intensity, sizeof(float));
    if (e == hipSuccess) e = hipMemset(u->max_intensity, 0, sizeof(float));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // (the null stream only: a resident open trace kernel of another unit is not waited for)
    if (e != hipSuccess) {
        rl_tonemap_unit_destroy(u);
        return fail(RL_E_HIP, std::string("tonemap unit allocation: ") + hipGetErrorString(e));
    }
    *out = u;
    return RL_OK;
}

int rl_tonemap_unit_destroy(RlTonemapUnit* u) {
    if (!u) return RL_OK;
    (void)hipSetDevice(u->device);
    if (u->rgb) (void)hipFree(u->rgb);
    if (u->srgb) (void)hipFree(u->srgb);
    if (u->max_intensity) (void)hipFree(u->max_intensity);
    delete u;
    return RL_OK;
}

int rl_tonemap_unit_tonemap(RlTonemapUnit* u, RlGatherUnit* gather) {
    if (!u || !gather) return fail(RL_E_INVALID, "null handle");
    if (gather->device != u->device || gather->width != u->width || gather->height != u->height)
        return fail(RL_E_STATE, "gather unit does not match the tonemap unit (device or size)");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    int cus = 256;
    if ((rc = cu_count_of(u->device, &cus)) != RL_OK) return rc;
    const uint32_t n_pixels = u->width * u->height;
    hipLaunchKernelGGL(rl_exposure_kernel, dim3(1), dim3(RL_EXPOSURE_BLOCK), 0, gather->stream, gather->acc, n_pixels, (float)n_pixels,
                       u->max_intensity);
    RL_HIP(hipGetLastError());
    hipLaunchKernelGGL(rl_tonemap_kernel, dim3(grid_for(n_pixels, cus)), dim3(RL_BLOCK), 0, gather->stream, gather->acc, n_pixels,
                       u->max_intensity, u->rgb, u->srgb);
    RL_HIP(hipGetLastError());
    u->last_stream = gather->stream;
    return RL_OK;
}

int rl_tonemap_unit_rgb(RlTonemapUnit* u, uint8_t* out) {
    if (!u || !out) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if (u->last_stream) RL_HIP(hipStreamSynchronize(u->last_stream));
    RL_HIP(hipMemcpy(out, u->rgb, (size_t)u->width * u->height * 3, hipMemcpyDeviceToHost));
    return RL_OK;
}

int rl_tonemap_unit_srgb_float(RlTonemapUnit* u, float* out, float* max_intensity) {
    if (!u) return fail(RL_E_INVALID, "null tonemap unit");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if (u->last_stream) RL_HIP(hipStreamSynchronize(u->last_stream));
    if (out) RL_HIP(hipMemcpy(out, u->srgb, (size_t)u->width * u->height * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (max_intensity) RL_HIP(hipMemcpy(max_intensity, u->max_intensity, sizeof(float), hipMemcpyDeviceToHost));
    return RL_OK;
}

// ---- GatherUnit-time exchange across GPUs (gather_unit.rs:49-64 with the plot buffers of G GPUs) ----

namespace {

// RCCL is bound with dlopen at the first rl_comm_* call, not at link time: the library then loads on hosts
// without RCCL (single-GPU use), and inside a process that already holds an RCCL (PyTorch ships its own copy
// under the same SONAME) the one copy in the process is shared instead of a second one being mapped.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr; // optional
    decltype(&ncclCommCount) CommCount = nullptr;   // optional: what the communicator itself says its size is
    std::string error, path;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL that belongs to the HIP runtime THIS library is running on: the one installed beside the
        // libamdhip64 that hipGetDeviceCount resolves to.  A process may hold two ROCm stacks (PyTorch bundles its own HIP,
        // HSA and RCCL under the same SONAMEs); picking RCCL by bare name can pair it with a second, uninitialised
        // HSA runtime ("no ROCm-capable device is detected").  Bare names are the fall-back.
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string& n : names) {
            api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (api.handle) {
                api.path = n;
                break;
            }
        }
        if (!api.handle) {
            api.error = std::string("RCCL (librccl.so.1) could not be loaded: ") + dlerror();
            return;
        }
#define RL_BIND(NAME)                                                                  \
    api.NAME = (decltype(api.NAME))dlsym(api.handle, "nccl" #NAME);                     \
    if (!api.NAME && api.error.empty()) api.error = "RCCL symbol nccl" #NAME " missing";
        RL_BIND(GetUniqueId) RL_BIND(CommInitRank) RL_BIND(CommInitAll) RL_BIND(CommDestroy) RL_BIND(Reduce)
        RL_BIND(GroupStart) RL_BIND(GroupEnd) RL_BIND(GetErrorString)
#undef RL_BIND
        api.GetVersion = (decltype(api.GetVersion))dlsym(api.handle, "ncclGetVersion");
        api.CommCount = (decltype(api.CommCount))dlsym(api.handle, "ncclCommCount");
    });
    return &api;
}

#define RL_NCCL(api, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) return fail(RL_E_HIP, std::string(#call) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

__global__ __launch_bounds__(RL_BLOCK) void rl_add_kernel(float* __restrict__ dst, const float* __restrict__ src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * RL_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RL_BLOCK) dst[i] = dst[i] + src[i];
}

} // namespace

int rl_comm_unique_id(uint8_t id[RL_COMM_ID_BYTES]) {
    if (!id) return fail(RL_E_INVALID, "null id");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    static_assert(sizeof(ncclUniqueId) == RL_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    RL_NCCL(api, api->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return RL_OK;
}

int rl_comm_init_rank(const uint8_t id[RL_COMM_ID_BYTES], int world, int rank, int device, RlComm** out) {
    if (!out) return fail(RL_E_INVALID, "null output handle");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(RL_E_INVALID, "bad communicator arguments");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t c = nullptr;
    RL_NCCL(api, api->CommInitRank(&c, world, u, rank));
    RlComm* r = new (std::nothrow) RlComm();
    if (!r) return fail(RL_E_INVALID, "out of host memory");
    r->device = device;
    r->rank = rank;
    r->world = world;
    r->nccl = c;
    *out = r;
    return RL_OK;
}

int rl_comm_init_all(const int* devices, int n, RlComm** out) {
    if (!devices || !out || n < 1) return fail(RL_E_INVALID, "bad communicator arguments");
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(RL_E_INVALID, "an RCCL communicator needs distinct devices (one rank per GPU)");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    int rc = use_device(devices[0]);
    if (rc != RL_OK) return rc;
    std::vector<ncclComm_t> comms(n, nullptr);
    RL_NCCL(api, api->CommInitAll(comms.data(), n, devices));
    for (int i = 0; i < n; ++i) {
        RlComm* r = new (std::nothrow) RlComm();
        if (!r) return fail(RL_E_INVALID, "out of host memory");
        r->device = devices[i];
        r->rank = i;
        r->world = n;
        r->nccl = comms[i];
        out[i] = r;
    }
    return RL_OK;
}

int rl_comm_destroy(RlComm* c) {
    if (!c) return RL_OK;
    RcclApi* api = rccl_api();
    (void)hipSetDevice(c->device);
    if (api->CommDestroy && c->nccl) (void)api->CommDestroy((ncclComm_t)c->nccl);
    delete c;
    return RL_OK;
}

int rl_comm_rank(const RlComm* c, int* rank, int* world) {
    if (!c) return fail(RL_E_INVALID, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return RL_OK;
}

int rl_comm_group_start(void) {
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    RL_NCCL(api, api->GroupStart());
    return RL_OK;
}

int rl_comm_group_end(void) {
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    RL_NCCL(api, api->GroupEnd());
    return RL_OK;
}

namespace {
// Reads the `n` oldest exchange timings of `u` into its totals (waits for them) and returns their events to the pool.
int exchanges_harvest(RlPlotUnit* u, size_t n) {
    if (n > u->exchanges.size()) n = u->exchanges.size();
    for (size_t i = 0; i < n; ++i) {
        EventPair& ep = u->exchanges[i];
        RL_HIP(hipEventSynchronize(ep.stop));
        float ms = 0.0f;
        RL_HIP(hipEventElapsedTime(&ms, ep.start, ep.stop));
        u->exchange_ms += (double)ms;
        u->exchange_count += 1;
        u->exchange_pool.push_back(ep);
    }
    u->exchanges.erase(u->exchanges.begin(), u->exchanges.begin() + (long)n);
    return RL_OK;
}
} // namespace

int rl_plot_unit_reduce(RlPlotUnit* u, RlComm* comm, int root) {
    if (!u || !comm) return fail(RL_E_INVALID, "null handle");
    if (comm->device != u->device) return fail(RL_E_STATE, "communicator rank and plot unit live on different devices");
    if (root < 0 || root >= comm->world) return fail(RL_E_INVALID, "root out of range");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(u)) != RL_OK) return rc; // a fused render begun into this buffer ends first
    const size_t count = (size_t)u->width * u->height * 3;
    // In place on the root; on the plot unit's stream, i.e. after every plot / fused splat into this buffer.
    // bracketed by events for rl_plot_unit_exchange_stats; a long run never asks, so the pairs are recycled and at most
    // 32 stay unread (the oldest is complete long before: it is on this unit's stream, 32 exchanges back)
    if (u->exchanges.size() >= 32 && (rc = exchanges_harvest(u, 16)) != RL_OK) return rc;
    EventPair ep;
    if (!u->exchange_pool.empty()) {
        ep = u->exchange_pool.back();
        u->exchange_pool.pop_back();
    } else {
        RL_HIP(hipEventCreate(&ep.start));
        RL_HIP(hipEventCreate(&ep.stop));
    }
    // (Inside rl_comm_group_start / _end -- one thread driving several ranks -- RCCL enqueues the collective at GroupEnd, so both
    // events land in front of it and the exchange reads as ~0 ms: rl_plot_unit_exchange_stats is meaningful for ungrouped
    // reduces only, which is what bench.py's one-rank-per-process runs issue.)
    hipError_t he = hipEventRecord(ep.start, u->stream);
    ncclResult_t ne = ncclSuccess;
    if (he == hipSuccess) ne = api->Reduce(u->xyz, u->xyz, count, ncclFloat32, ncclSum, root, (ncclComm_t)comm->nccl, u->stream);
    if (he == hipSuccess && ne == ncclSuccess) he = hipEventRecord(ep.stop, u->stream);
    if (he != hipSuccess || ne != ncclSuccess) {
        u->exchange_pool.push_back(ep); // not leaked
        if (ne != ncclSuccess) return fail(RL_E_HIP, std::string("ncclReduce: ") + api->GetErrorString(ne));
        return fail(RL_E_HIP, std::string("hipEventRecord: ") + hipGetErrorString(he));
    }
    u->exchanges.push_back(ep);
    return RL_OK;
}

int rl_plot_unit_exchange_stats(RlPlotUnit* u, uint64_t* exchanges, double* device_ms) {
    if (!u) return fail(RL_E_INVALID, "null plot unit");
    int rc = use_device(u->device);
    if (rc != RL_OK) return rc;
    if ((rc = exchanges_harvest(u, u->exchanges.size())) != RL_OK) return rc;
    if (exchanges) *exchanges = u->exchange_count;
    if (device_ms) *device_ms = u->exchange_ms;
    return RL_OK;
}

int rl_comm_info(const RlComm* c, int* rank, int* world, int* rccl_version, char* library_path, uint32_t path_cap) {
    if (!c) return fail(RL_E_INVALID, "null communicator");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(RL_E_NO_DEVICE, api->error);
    if (rank) *rank = c->rank;
    if (world) { // the communicator's own count where RCCL exports it: "did RCCL see N ranks" answered by RCCL
        int n = c->world;
        if (api->CommCount) RL_NCCL(api, api->CommCount((ncclComm_t)c->nccl, &n));
        *world = n;
    }
    if (rccl_version) {
        int v = 0;
        if (api->GetVersion) RL_NCCL(api, api->GetVersion(&v));
        *rccl_version = v;
    }
    if (library_path && path_cap) {
        std::snprintf(library_path, path_cap, "%s", api->path.c_str());
    }
    return RL_OK;
}

int rl_plot_unit_add(RlPlotUnit* dst, RlPlotUnit* src) {
    if (!dst || !src) return fail(RL_E_INVALID, "null handle");
    if (dst == src) return fail(RL_E_INVALID, "a plot unit cannot be added to itself");
    if (dst->device != src->device || dst->width != src->width || dst->height != src->height)
        return fail(RL_E_STATE, "plot units do not match (device or size)");
    int rc = use_device(dst->device);
    if (rc != RL_OK) return rc;
    if ((rc = plot_settle(dst)) != RL_OK || (rc = plot_settle(src)) != RL_OK) return rc; // begun fused renders end first
    int cus = 256;
    if ((rc = cu_count_of(dst->device, &cus)) != RL_OK) return rc;
    const uint64_t n = (uint64_t)dst->width * dst->height * 3;
    RL_HIP(hipEventRecord(src->ready, src->stream));
    RL_HIP(hipStreamWaitEvent(dst->stream, src->ready, 0));
    hipLaunchKernelGGL(rl_add_kernel, dim3(grid_for(n, cus)), dim3(RL_BLOCK), 0, dst->stream, dst->xyz, src->xyz, n);
    RL_HIP(hipGetLastError());
    RL_HIP(hipEventRecord(dst->plotted, dst->stream));
    RL_HIP(hipStreamWaitEvent(src->stream, dst->plotted, 0)); // src may be cleared or splatted into only after it was read
    return RL_OK;
}

int rl_gather_unit_allreduce(RlGatherUnit* gather, RlPlotUnit* plot, RlComm* comm) {
    if (!plot || !comm) return fail(RL_E_INVALID, "null handle");
    if (comm->rank == 0 && !gather) return fail(RL_E_INVALID, "rank 0 accumulates: it needs the gather unit");
    int rc = rl_plot_unit_reduce(plot, comm, 0);
    if (rc != RL_OK) return rc;
    if (comm->rank == 0) return rl_gather_unit_accumulate(gather, plot); // Kahan + clear (gather_unit.rs:49-64, app.rs:147)
    return rl_plot_unit_clear(plot);
}

// Diagnostics: how many launches carried k blocking render calls (k = 1 .. 256) on `device` since the library was
// loaded; out holds 257 counters.
int rl_debug_batch_histogram(int device, uint64_t* out) {
    if (!out) return fail(RL_E_INVALID, "null argument");
    for (uint32_t k = 0; k <= RL_OPEN_CAP; ++k) out[k] = 0;
    int rc = use_device(device);
    if (rc == RL_OK) rc = sessions_quiesce(device); // the open launches still running end first: then they are counted
    if (rc != RL_OK) return rc;
    DeviceSessions* d = sessions_of(device);
    std::lock_guard<std::mutex> guard(d->lock);
    for (uint32_t k = 0; k <= RL_OPEN_CAP; ++k) out[k] += d->histogram[k];
    if (getenv("RL_OPEN_LAUNCH_TIMING") && d->calls)
        fprintf(stderr, "open launches on device %d: %llu calls, mean us per call: wait for the target %.1f, admission %.1f, completion %.1f\n",
                device, (unsigned long long)d->calls, d->presync_us / d->calls, d->admit_us / d->calls, d->wait_us / d->calls);
    d->presync_us = d->admit_us = d->wait_us = 0.0;
    d->calls = 0;
    return RL_OK;
}

// Diagnostics: launches of each of the 24 instantiations of the trace kernel since the library was loaded; index = 8 * (the whole
// scene staged in LDS) + 4 * (fused with the splat) + 2 * (open launch) + 1 * (prisms with a second bound), and 16 + the low three
// bits for the instantiations that stage the tables only (trace_kernel_variant).
int rl_debug_variant_launches(uint64_t* out) {
    if (!out) return fail(RL_E_INVALID, "null argument");
    for (int k = 0; k < 24; ++k) out[k] = g_variant_launches[k].load(std::memory_order_relaxed);
    return RL_OK;
}

// ---- device-side math probe (tests) -------------------------------------------------------------

// Not part of the reference's interface: evaluates csrc/rl_math.h on the GPU so the tests can
// verify that hipcc and g++ agree bit-for-bit.  fn as in rl_math_probe_kernel.
int rl_debug_math_probe(int device, int fn, const float* x, float* y, uint32_t n) {
    if (!x || !y || n == 0) return fail(RL_E_INVALID, "null argument");
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    float *dx = nullptr, *dy = nullptr;
    RL_HIP(hipMalloc((void**)&dx, n * sizeof(float)));
    hipError_t e = hipMalloc((void**)&dy, n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(dx, x, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rl_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, fn, dx, dy, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(y, dy, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (e != hipSuccess) return fail(RL_E_HIP, std::string("math probe: ") + hipGetErrorString(e));
    return RL_OK;
}

// The short forms (fn 16 rl_sqrtf, 17 rl_recipf, 18 rl_div200f) against the compiler's IEEE expansions for every float with bits in
// [lo_bits, hi_bits) -- and its negative when both_signs -- on the device.  counts[0] = arguments that differ, counts[1] = arguments
// compared; *example = one argument that differs (bits), if any.
int rl_debug_math_sweep(int device, int fn, uint32_t lo_bits, uint32_t hi_bits, int both_signs, uint64_t* counts, uint32_t* example) {
    if (!counts || !example || fn < 16 || fn > 18 || hi_bits <= lo_bits) return fail(RL_E_INVALID, "math sweep: fn 16..18, lo < hi");
    int rc = use_device(device);
    if (rc != RL_OK) return rc;
    unsigned long long* bad = nullptr;
    RL_HIP(hipMalloc((void**)&bad, 32));
    hipError_t e = hipMemset(bad, 0, 32);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rl_math_sweep_kernel, dim3(8192), dim3(RL_BLOCK), 0, 0, fn, lo_bits, hi_bits, both_signs, bad, (uint32_t*)(bad + 2));
        e = hipGetLastError();
    }
    unsigned long long host[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(host, bad, 32, hipMemcpyDeviceToHost);
    (void)hipFree(bad);
    if (e != hipSuccess) return fail(RL_E_HIP, std::string("math sweep: ") + hipGetErrorString(e));
    counts[0] = host[0];
    counts[1] = host[1];
    *example = (uint32_t)host[2];
    return RL_OK;
}

int rl_debug_prism_count(const RlScene* scene, uint32_t* n_prisms) {
    if (!scene || !n_prisms) return fail(RL_E_INVALID, "null argument");
    *n_prisms = scene->lay.n_prisms;
    return RL_OK;
}

int rl_debug_prism_probe(const RlScene* scene, uint32_t prism, const float* rays, uint32_t n, uint32_t* out) {
    if (!scene || !rays || !out) return fail(RL_E_INVALID, "null argument");
    if (prism >= scene->lay.n_prisms) return fail(RL_E_INVALID, "no such prism");
    if (n == 0) return RL_OK;
    int rc = use_device(scene->device);
    if (rc != RL_OK) return rc;
    float* dr = nullptr;
    uint32_t* dout = nullptr;
    hipError_t e = hipMalloc((void**)&dr, (size_t)n * 6 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&dout, (size_t)n * 5 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(dr, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rl_prism_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0,
                           scene->blob + scene->lay.off_prisms + (size_t)RL_PRISM_STRIDE * prism, dr, n, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 5 * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (dr) (void)hipFree(dr);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(RL_E_HIP, std::string("prism probe: ") + hipGetErrorString(e));
    return RL_OK;
}

#ifdef RL_STATS
// Diagnostic build only (not in the header): copies and clears the trace kernel's event counters.
int rl_stats_read(unsigned long long* out, int n) {
    unsigned long long host[48];
    if (hipDeviceSynchronize() != hipSuccess) return RL_E_HIP;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(rl_stat_counters), sizeof host) != hipSuccess) return RL_E_HIP;
    for (int i = 0; i < n && i < 48; ++i) out[i] = host[i];
    std::memset(host, 0, sizeof host);
    if (hipMemcpyToSymbol(HIP_SYMBOL(rl_stat_counters), host, sizeof host) != hipSuccess) return RL_E_HIP;
    return RL_OK;
}
#endif

} // extern "C"
