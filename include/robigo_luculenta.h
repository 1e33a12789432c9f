/* robigo_luculenta.h -- C ABI of the MI355X-native hot path of robigo-luculenta.
 *
 * The reference has no FFI; its seam is the `Task` enum and the four unit structs that worker
 * threads execute (app.rs:113-164).  Each entry point below replaces one unit method (cited), so a
 * Rust host keeps task_scheduler.rs untouched and swaps the bodies of App::execute_*_task for these
 * calls (INTEGRATION.md shows the `extern "C"` block and the #[repr(C)] structs).
 *
 * Conventions: every function returns 0 on success or a negative RL_E_* code; the message is
 * available through rl_last_error() (thread-local).  Nothing throws or aborts across the boundary
 * (the reference panics instead: app.rs:107,163; gather_unit.rs:69-70).  Handles are opaque, own
 * their device buffers, and are used by one host thread at a time -- the same exclusive ownership
 * the reference gets by moving Box<Unit> through Task (task_scheduler.rs:26-41).  A scene handle is
 * immutable after creation and may be shared (Arc<Scene>, app.rs:63).
 *
 * There is no CPU implementation behind this ABI: every compute entry point runs hand-written
 * gfx950 kernels and fails with RL_E_NO_DEVICE when no GPU is present.
 */
#ifndef ROBIGO_LUCULENTA_H
#define ROBIGO_LUCULENTA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- frozen data layouts ------------------------------------------------------------------ */

/* vector3.rs:20-25 (12 bytes; gather_unit.rs:75 relies on that size). */
typedef struct RlVector3 {
    float x, y, z;
} RlVector3;

/* trace_unit.rs:23-37. */
typedef struct RlMappedPhoton {
    float x, y, probability, wavelength;
} RlMappedPhoton;

/* Surfaces (geometry.rs) and materials (material.rs) as plain data.  The reference hard-codes its
 * scene in Rust (app.rs:166-363); here a scene is an array of RlObjectDesc in scan order. */
enum RlSurfaceKind {
    RL_SURFACE_SPHERE = 0,     /* Sphere::new(position = v0, radius = f0)              geometry.rs:195-200 */
    RL_SURFACE_PLANE = 1,      /* Plane::new(normal = v0, offset = v1)                 geometry.rs:44-51   */
    RL_SURFACE_CIRCLE = 2,     /* Circle::new(normal = v0, position = v1, radius = f0) geometry.rs:142-150 */
    RL_SURFACE_PARABOLOID = 3, /* Paraboloid::new(normal = v0, offset = v1, focal_distance = f0) :286-295  */
    RL_SURFACE_HEX_PRISM = 4   /* new_hexagonal_prism(axis = v0, offset = v1, edge_length = f0,
                                  bevel_size = f1, angle = f2, height = f3)            geometry.rs:493-515 */
};

enum RlMaterialKind {
    RL_MATERIAL_BLACK_BODY = 0,       /* emissive; BlackBodyMaterial::new(kelvins = m0, intensity = m1) material.rs:93-99 */
    RL_MATERIAL_DIFFUSE_GREY = 1,     /* DiffuseGreyMaterial::new(reflectance = m0)             material.rs:115-119 */
    RL_MATERIAL_DIFFUSE_COLOURED = 2, /* DiffuseColouredMaterial::new(refl = m0, wavel = m1, dev = m2)  :145-152 */
    RL_MATERIAL_GLOSSY_MIRROR = 3,    /* GlossyMirrorMaterial::new(gloss = m0)                  material.rs:177-182 */
    RL_MATERIAL_SF10_GLASS = 4,       /* Sf10GlassMaterial                                      material.rs:199 */
    RL_MATERIAL_SOAP_BUBBLE = 5       /* SoapBubbleMaterial                                     material.rs:265 */
};

typedef struct RlObjectDesc {
    uint32_t surface_kind;  /* enum RlSurfaceKind */
    uint32_t material_kind; /* enum RlMaterialKind */
    RlVector3 v0, v1;       /* surface constructor vectors, see RlSurfaceKind */
    float f0, f1, f2, f3;   /* surface constructor scalars */
    float m0, m1, m2;       /* material constructor scalars, see RlMaterialKind */
} RlObjectDesc;

/* The camera as a function of time t in [0,1]; parametrises make_camera (app.rs:327-357):
 *   phi = PI * (phi0 + phi1 * t);  alpha = PI * (alpha0 + alpha1 * t);  distance = dist0 + dist1 * t
 *   position = (cos(alpha) sin(phi), cos(alpha) cos(phi), sin(alpha)) * distance
 *   orientation = rotation((0,0,-1), phi + PI) * rotation((1,0,0), -alpha)
 *   field_of_view = PI * fov_over_pi;  focal_distance = distance * focal_factor */
typedef struct RlCameraDesc {
    float phi0, phi1;
    float alpha0, alpha1;
    float dist0, dist1;
    float fov_over_pi;
    float focal_factor;
    float depth_of_field;
    float chromatic_abberation;
} RlCameraDesc;

typedef struct RlSceneDesc {
    uint32_t n_objects;
    const RlObjectDesc* objects;
    RlCameraDesc camera;
} RlSceneDesc;

/* Built-in scene generators. */
enum RlBuiltinScene {
    RL_SCENE_DEMO = 0,         /* App::set_up_scene, app.rs:166-363; param = seeds (0 -> 100 as in app.rs:238) */
    RL_SCENE_GLASS_STRESS = 1, /* fixed objects 0-6 + three rings of SF10 prisms (BASELINE config 3); param unused */
};

/* Where the trace kernel reads the primitive list from. */
// RL_FETCH_LDS stages as much as fits beside the waves' scratch: the whole scene (up to about 1,300 objects), else its tables
// (planes, paraboloids, prisms, the cull table) with the spheres and objects read from L2 / HBM, else -- tables beyond ~37 KB,
// tens of thousands of objects -- nothing.  RL_FETCH_GLOBAL never stages anything.
enum RlPrimitiveFetch {
    RL_FETCH_LDS = 0,   /* primitives + CIE tables staged in LDS by each workgroup */
    RL_FETCH_GLOBAL = 1 /* wave-uniform loads from HBM/L2 through the scalar cache */
};

enum RlError {
    RL_OK = 0,
    RL_E_INVALID = -1,   /* bad argument */
    RL_E_NO_DEVICE = -2, /* no gfx950 device / HIP runtime failure at start-up */
    RL_E_HIP = -3,       /* a HIP call failed; see rl_last_error() */
    RL_E_IO = -4,        /* checkpoint file could not be opened / written */
    RL_E_STATE = -5      /* handle used out of protocol (e.g. size mismatch between units) */
};

typedef struct RlScene RlScene;
typedef struct RlTraceUnit RlTraceUnit;
typedef struct RlPlotUnit RlPlotUnit;
typedef struct RlGatherUnit RlGatherUnit;
typedef struct RlTonemapUnit RlTonemapUnit;
typedef struct RlScheduler RlScheduler;
typedef struct RlComm RlComm;

const char* rl_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int rl_device_count(void);
/* The PCI bus id ("0000:c1:00.0") of visible device `device`: what tells two processes whether "device 0" is the same GPU
 * for both (ranks launched with a per-rank device mask) or not -- RCCL admits one rank per GPU. */
int rl_device_pci_bus_id(int device, char* out, uint32_t cap);
const char* rl_version(void);
/* 16 hex digits: a hash of the sources this library's device code was compiled from (csrc/Makefile).  Profiles
 * record it so that counter-derived figures are never quoted for a different build. */
const char* rl_build_id(void);

/* ---- scene (scene.rs:23-35, app.rs:166-363) ------------------------------------------------ */

/* Fills `objects` (capacity `cap`) with the built-in scene and writes the object count to
 * *n_objects and the camera to *camera.  Pure host code, needs no device.  When cap is too small
 * returns RL_E_INVALID with *n_objects set to the required count. */
int rl_scene_builtin_desc(int which, int param, RlObjectDesc* objects, uint32_t cap, uint32_t* n_objects,
                          RlCameraDesc* camera);
/* Scene descriptions as files (the reference hard-codes its scene, app.rs:166-363).  Little-endian binary:
 * "RLSC" magic, u32 version = 1, u32 n_objects, RlCameraDesc (40 bytes), n_objects x RlObjectDesc (60 bytes).
 * Host-only.  rl_scene_desc_load follows rl_scene_builtin_desc's capacity protocol. */
int rl_scene_desc_save(const char* path, const RlSceneDesc* desc);
int rl_scene_desc_load(const char* path, RlObjectDesc* objects, uint32_t cap, uint32_t* n_objects, RlCameraDesc* camera);
/* Flattens the description (hex prisms become 8 half-spaces, paraboloids get their derived
 * fields, black bodies their normalisation factor) and uploads it to `device`.  Replaces
 * App::set_up_scene + Arc::new (app.rs:63). */
int rl_scene_create(const RlSceneDesc* desc, int device, RlScene** out);
int rl_scene_destroy(RlScene* scene);

/* ---- TraceUnit (trace_unit.rs:51-168) ------------------------------------------------------ */

/* TraceUnit::new(id, width, height) (trace_unit.rs:64-77); n_photons is the batch size the
 * reference fixes at 1024*512 (trace_unit.rs:67). */
int rl_trace_unit_create(int device, uint32_t id, uint32_t width, uint32_t height, uint32_t n_photons,
                         RlTraceUnit** out);
int rl_trace_unit_destroy(RlTraceUnit* unit);
/* Selects RL_FETCH_LDS (default) or RL_FETCH_GLOBAL for subsequent renders. */
int rl_trace_unit_set_fetch(RlTraceUnit* unit, int primitive_fetch);
/* TraceUnit::render(&mut self, &Scene) (trace_unit.rs:151-168): fills the unit's mapped_photons.
 * Photon i of this call is path (first_path_index + i) of RNG stream `stream` under `seed`.
 * Complete on return: mapped_photons may be plotted or downloaded.  Calls made from several threads (the reference's
 * workers, app.rs:92-134) on units of one device with the same scene, seed, stream and image size share OPEN
 * LAUNCHES: a call whose batch is a multiple of 64 paths is appended to a trace kernel that is already running on
 * the device if there is one (otherwise it starts one), and returns as soon as ITS paths are finished while the
 * kernel goes on with the other callers' -- so nothing is launched per call and the drain tail of one batch overlaps
 * the next batches.  The results are bit-identical to separate launches.  Path and segment counters are kept per
 * call; the run time of an open launch is split among the units whose calls it carried, by paths, when it has ended. */
int rl_trace_unit_render(RlTraceUnit* unit, const RlScene* scene, uint64_t seed, uint32_t stream,
                         uint64_t first_path_index);
/* rl_trace_unit_render in two halves, for a host thread that has something else to do meanwhile (feeding other GPUs,
 * taking the next task: rl_app_run does both): _begin appends the call to the device's open launch (or starts one) and
 * returns at once, _end waits until the call's paths are finished (a no-op without a begun render).  One begun render
 * per unit at a time.  Everything that reads mapped_photons ends a begun render by itself: rl_plot_unit_plot (for the
 * units it plots), rl_trace_unit_photons, rl_trace_unit_sync, rl_trace_unit_stats, rl_trace_unit_destroy.  The thread
 * that ends a render need not be the one that began it, as long as the unit changes hands the way every unit must
 * (one user at a time, handed over through a lock or a channel -- the reference's Task does that).
 * A device runs at most FOUR open launches at a time, one per (scene, seed, stream, image size, fetch mode, fused or not)
 * combination in use.  A render begun for a fifth combination while begun-and-not-ended renders hold all four is not
 * refused and does not wait: it gets a plain launch of its own on the unit's stream (same results; _end waits for it). */
int rl_trace_unit_render_begin(RlTraceUnit* unit, const RlScene* scene, uint64_t seed, uint32_t stream,
                               uint64_t first_path_index);
int rl_trace_unit_render_end(RlTraceUnit* unit);
/* The same without the final wait: the launch is queued on the unit's stream and rl_trace_unit_sync() (or
 * rl_plot_unit_plot, which orders itself after it on the device) completes it.  Lets one host thread start the
 * same task on several GPUs before waiting for any of them. */
int rl_trace_unit_render_async(RlTraceUnit* unit, const RlScene* scene, uint64_t seed, uint32_t stream,
                               uint64_t first_path_index);
/* Fused TraceUnit::render + PlotUnit::plot (trace_unit.rs:151-168 + plot_unit.rs:87-95): traces
 * n_paths paths (any count, not limited to the unit's batch size) and splats every non-zero
 * contribution straight into `plot` with f32 atomics; mapped_photons is not written.  After this a
 * rl_plot_unit_plot for the same photons must NOT be issued.  Asynchronous on the unit's stream;
 * rl_trace_unit_sync() or any download waits for it. */
int rl_trace_unit_render_fused(RlTraceUnit* unit, const RlScene* scene, RlPlotUnit* plot, uint64_t seed,
                               uint32_t stream, uint64_t first_path_index, uint64_t n_paths);
/* The same, complete on return, for hosts whose workers wait for their task anyway (the reference's do): calls on
 * units of one device (same scene, seed, stream, image size; n_paths a multiple of 64) share open launches like
 * rl_trace_unit_render's, each splatting into its own call's plot unit.  Results are those of separate calls up to
 * the order of the float atomics. */
int rl_trace_unit_render_fused_sync(RlTraceUnit* unit, const RlScene* scene, RlPlotUnit* plot, uint64_t seed,
                                    uint32_t stream, uint64_t first_path_index, uint64_t n_paths);
/* Its first half.  The begun render belongs to `plot` -- the trace unit is free for the next call, on any thread, at
 * once and may even be destroyed -- and is ended by whatever uses the plot unit's buffer next: rl_plot_unit_sync,
 * rl_gather_unit_accumulate / _allreduce, rl_plot_unit_reduce / _add / _plot / _clear / _download / _upload /
 * _device_buffer / _destroy, or another render begun into it.  Its paths and segments appear in the trace unit's
 * rl_trace_unit_stats once it has ended. */
int rl_trace_unit_render_fused_begin(RlTraceUnit* unit, const RlScene* scene, RlPlotUnit* plot, uint64_t seed,
                                     uint32_t stream, uint64_t first_path_index, uint64_t n_paths);
int rl_trace_unit_sync(RlTraceUnit* unit);
/* Copies mapped_photons (trace_unit.rs:56) to host memory; `out` holds n_photons entries. */
int rl_trace_unit_photons(RlTraceUnit* unit, RlMappedPhoton* out);
/* Cumulative counters since creation: paths traced and path segments (= Scene::intersect calls,
 * scene.rs:39) and the device time spent in trace kernels in milliseconds. */
int rl_trace_unit_stats(RlTraceUnit* unit, uint64_t* paths, uint64_t* segments, double* kernel_ms);

/* ---- PlotUnit (plot_unit.rs:23-102) -------------------------------------------------------- */

/* PlotUnit::new(id, width, height) (plot_unit.rs:43-52).  If external_xyz is non-NULL it must be
 * a device pointer to width*height*3 floats on `device` that outlives the unit (lets the caller
 * run a collective on the buffer); otherwise the unit allocates and zeroes its own. */
int rl_plot_unit_create(int device, uint32_t id, uint32_t width, uint32_t height, float* external_xyz,
                        RlPlotUnit** out);
int rl_plot_unit_destroy(RlPlotUnit* unit);
/* PlotUnit::plot(&mut self, &[MappedPhoton]) for each given trace unit (app.rs:136-141).  ASYNCHRONOUS: the
 * splat kernels are queued on the plot unit's own (non-blocking) stream, after the trace units' renders; the
 * trace units may be rendered again at once (their next render waits for this plot on the device).  A later
 * rl_gather_unit_accumulate / rl_plot_unit_reduce / rl_plot_unit_download of this unit is ordered after it;
 * anything else that reads the buffer (rl_plot_unit_device_buffer) must call rl_plot_unit_sync first. */
int rl_plot_unit_plot(RlPlotUnit* unit, RlTraceUnit* const* trace_units, uint32_t n_trace_units);
/* PlotUnit::clear (plot_unit.rs:98-102); queued on the unit's stream. */
int rl_plot_unit_clear(RlPlotUnit* unit);
/* Blocks until everything queued into this plot unit so far (plots, fused renders, exchanges, clears) is done. */
int rl_plot_unit_sync(RlPlotUnit* unit);
/* Device pointer of tristimulus_buffer (plot_unit.rs:35): width*height RlVector3, row-major. */
int rl_plot_unit_device_buffer(RlPlotUnit* unit, float** device_xyz);
int rl_plot_unit_download(RlPlotUnit* unit, RlVector3* out);
/* Overwrites tristimulus_buffer from host memory (after everything queued into the unit): the host-staged form
 * of the exchange, for ranks that share a GPU, and the tests. */
int rl_plot_unit_upload(RlPlotUnit* unit, const RlVector3* in);

/* ---- GatherUnit (gather_unit.rs:24-92) ----------------------------------------------------- */

/* GatherUnit::new(width, height) WITHOUT the implicit read() of ./buffer.raw
 * (gather_unit.rs:42-43): resuming is explicit through rl_gather_unit_load. */
int rl_gather_unit_create(int device, uint32_t width, uint32_t height, RlGatherUnit** out);
int rl_gather_unit_destroy(RlGatherUnit* unit);
/* GatherUnit::accumulate(&plot.tristimulus_buffer) followed by plot.clear()
 * (app.rs:143-148, gather_unit.rs:49-64): Kahan-compensated, never re-associated. */
int rl_gather_unit_accumulate(RlGatherUnit* unit, RlPlotUnit* plot);
/* GatherUnit::save / read (gather_unit.rs:68-92): headerless tristimulus then compensation
 * buffers, 12 native-endian bytes per pixel each.  A short file leaves the tail untouched
 * (read.rs:20-32). */
int rl_gather_unit_save(RlGatherUnit* unit, const char* path);
int rl_gather_unit_load(RlGatherUnit* unit, const char* path);
int rl_gather_unit_download(RlGatherUnit* unit, RlVector3* tristimulus, RlVector3* compensation);
/* Blocks until every accumulate / tonemap queued on this gather unit is done (they run on its own stream). */
int rl_gather_unit_sync(RlGatherUnit* unit);

/* ---- the GatherUnit-time exchange between GPUs (no reference counterpart: the reference is one process on
 * one CPU) ---------------------------------------------------------------------------------------
 * Samples shard: every GPU renders the full frame with its own RNG stream into its own plot units, and the only
 * exchange is the element-wise f32 sum of the tristimulus buffers (3*W*H floats) onto rank 0 when a plot unit
 * is gathered -- one ncclReduce over xGMI -- followed by rank 0's Kahan accumulation (gather_unit.rs:49-64).
 * RCCL is loaded at run time (dlopen) by the first rl_comm_* call; RL_E_NO_DEVICE if it is absent. */
#define RL_COMM_ID_BYTES 128
/* One rank per process (torchrun / MPI style launchers): rank 0 obtains an id, hands the 128 bytes to the other
 * ranks over any channel, and every rank joins with its own device. */
int rl_comm_unique_id(uint8_t id[RL_COMM_ID_BYTES]);
int rl_comm_init_rank(const uint8_t id[RL_COMM_ID_BYTES], int world, int rank, int device, RlComm** out);
/* One process driving n distinct GPUs: out[i] is rank i on devices[i]. */
int rl_comm_init_all(const int* devices, int n, RlComm** out);
int rl_comm_destroy(RlComm* comm);
int rl_comm_rank(const RlComm* comm, int* rank, int* world);
/* What the exchange really runs on, for a measurement that has to explain itself: the rank, the size RCCL reports for the
 * communicator (ncclCommCount), the RCCL version (ncclGetVersion, e.g. 22707) and the path of the library that was
 * loaded.  Any output may be NULL. */
int rl_comm_info(const RlComm* comm, int* rank, int* world, int* rccl_version, char* library_path, uint32_t path_cap);
/* ncclGroupStart / ncclGroupEnd: a single host thread that issues the collective for several ranks brackets
 * the calls with these (one thread or process per rank needs neither). */
int rl_comm_group_start(void);
int rl_comm_group_end(void);
/* The exchange: sums this plot unit's tristimulus buffer with those of the other ranks into rank `root`'s
 * buffer, in place (ncclReduce, f32, sum).  Collective: every rank calls it with its own plot unit of the same
 * size.  Queued on the plot unit's stream, i.e. after every plot / fused render into the buffer. */
int rl_plot_unit_reduce(RlPlotUnit* unit, RlComm* comm, int root);
/* Device time of this unit's exchanges so far: every rl_plot_unit_reduce is bracketed by events on the plot unit's
 * stream.  Waits for the ones still in flight.  Cumulative since creation. */
int rl_plot_unit_exchange_stats(RlPlotUnit* unit, uint64_t* exchanges, double* device_ms);
/* dst += src for two plot units on the SAME device (two RNG streams rendered by one GPU); src is left as it is. */
int rl_plot_unit_add(RlPlotUnit* dst, RlPlotUnit* src);
/* Task::Gather on G GPUs in one call per rank: rl_plot_unit_reduce(plot, comm, 0), then on rank 0
 * rl_gather_unit_accumulate(gather, plot) and on the other ranks rl_plot_unit_clear(plot) (gather may be NULL
 * there).  For one thread or process per rank; a single thread driving several ranks uses the three calls
 * itself with the reduces inside rl_comm_group_start/end. */
int rl_gather_unit_allreduce(RlGatherUnit* gather, RlPlotUnit* plot, RlComm* comm);

/* ---- TonemapUnit (tonemap_unit.rs:21-100, srgb.rs:20-41) ------------------------------------ */

int rl_tonemap_unit_create(int device, uint32_t width, uint32_t height, RlTonemapUnit** out);
int rl_tonemap_unit_destroy(RlTonemapUnit* unit);
/* TonemapUnit::tonemap(&gather.tristimulus_buffer) (tonemap_unit.rs:73-100). */
int rl_tonemap_unit_tonemap(RlTonemapUnit* unit, RlGatherUnit* gather);
/* rgb_buffer (tonemap_unit.rs:30): width*height*3 bytes RGB8. */
int rl_tonemap_unit_rgb(RlTonemapUnit* unit, uint8_t* out);
/* The clamped sRGB value before `* 255 as u8` (tonemap_unit.rs:88-98), width*height*3 floats, and
 * the exposure estimate of find_exposure (tonemap_unit.rs:55-69).  Either pointer may be NULL. */
int rl_tonemap_unit_srgb_float(RlTonemapUnit* unit, float* out, float* max_intensity);

/* ---- Task / TaskScheduler (task_scheduler.rs:26-182) --------------------------------------- */

enum RlTaskKind { RL_TASK_SLEEP = 0, RL_TASK_TRACE = 1, RL_TASK_PLOT = 2, RL_TASK_GATHER = 3, RL_TASK_TONEMAP = 4 };

/* Capacity of RlTask::units: 3 * concurrency trace units must fit, so at most 85 worker threads (the reference
 * uses num_cpus::get(), app.rs:55; a GPU is saturated by 8-16 workers).  rl_scheduler_create and
 * rl_app_run return RL_E_INVALID with a message naming the limit beyond it. */
#define RL_TASK_MAX_UNITS 256

/* enum Task by value (task_scheduler.rs:26-41), units named by their ids (trace_unit.rs:59,
 * plot_unit.rs:37).  unit = the trace/plot unit of Trace/Plot; units[] = the trace units of a Plot
 * or the plot units of a Gather. */
typedef struct RlTask {
    uint32_t kind; /* enum RlTaskKind */
    uint32_t unit;
    uint32_t n_units;
    uint32_t units[RL_TASK_MAX_UNITS];
} RlTask;

/* TaskScheduler::new(concurrency, width, height) (task_scheduler.rs:91-125) over unit ids only:
 * 3*concurrency trace units, max(1, concurrency/2) plot units.  tonemap_interval_ms replaces the
 * hard-coded 30 s (task_scheduler.rs:44-46). */
int rl_scheduler_create(uint32_t concurrency, int64_t tonemap_interval_ms, RlScheduler** out);
int rl_scheduler_destroy(RlScheduler* s);
/* TaskScheduler::get_new_task(completed) (task_scheduler.rs:127-182).  now_ms is the caller's
 * monotonic clock (the reference calls time::get_time() inside). */
int rl_scheduler_get_new_task(RlScheduler* s, const RlTask* completed, int64_t now_ms, RlTask* next);
/* Mean and standard deviation of batches/sec over the last <= 512 tonemap intervals
 * (task_scheduler.rs:308-325). */
int rl_scheduler_performance(RlScheduler* s, float* mean, float* stddev);

/* ---- App (app.rs:48-164) ----------------------------------------------------------------------- */

typedef struct RlAppConfig {
    uint32_t width, height;      /* main.rs:47-48 (1280 x 720 there) */
    int device;                  /* GPU that takes the place of the CPU worker pool */
    uint32_t concurrency;        /* scheduler depth = worker threads unless `threads` says otherwise; the reference uses num_cpus::get() (app.rs:55) */
    uint32_t photons_per_batch;  /* 0 -> 1024*512 (trace_unit.rs:67) */
    uint64_t seed;
    uint32_t stream;             /* RNG stream (multi-GPU: the rank) */
    int builtin_scene;           /* enum RlBuiltinScene */
    int builtin_param;
    uint64_t max_batches;        /* stop after this many trace tasks (the reference never stops, main.rs:57) */
    int64_t tonemap_interval_ms; /* 30000 in the reference (task_scheduler.rs:44-46) */
    int fused;                   /* 0: Trace fills mapped_photons, Plot splats them (reference structure);
                                    1: Trace renders straight into the plot unit it will be plotted by */
    const char* output_ppm;      /* image written after every tonemap: "*.png" -> PNG (the reference's output.png,
                                    main.rs:61), anything else -> binary PPM (P6); NULL = none */
    const char* checkpoint;      /* GatherUnit::save target, written at every tonemap and at the end; NULL = none */
    int resume;                  /* non-zero: rl_gather_unit_load(checkpoint) before rendering (gather_unit.rs:42-43) */
    int verbose;                 /* print the reference's progress lines (task_scheduler.rs:242-325) */
    uint32_t sleep_us;           /* how long Task::Sleep waits before the worker asks again.  The reference sleeps
                                    100 ms (app.rs:129) because its tasks take seconds; a task here takes about a
                                    millisecond, so 0 selects 200 us.  100000 restores the reference's value. */
    uint64_t first_batch;        /* index of the first batch this run renders; batch b is the path indices
                                    [b * photons_per_batch, (b + 1) * photons_per_batch) of every rank's RNG stream.
                                    With `resume` the larger of this and the index stored beside the checkpoint
                                    ("<checkpoint>.next", written with every save) is used, so a resumed run adds new
                                    samples instead of repeating the old ones.  RlAppStats::next_batch continues a run
                                    by hand. */
    uint32_t n_devices;          /* 0 or 1: render on `device`.  G > 1: one process drives G ranks; every scheduler
                                    unit is one unit per rank, rank r uses RNG stream `stream + r`, and Task::Gather
                                    sums the ranks' plot buffers onto rank 0 first (rl_plot_unit_reduce over xGMI for
                                    distinct GPUs, rl_plot_unit_add for ranks that share one) */
    int blocking_trace;          /* 0 (default): a task BEGINS its render (rl_trace_unit_render_begin, un-fused in the Trace task;
                                    rl_trace_unit_render_fused_begin, fused in the Plot task) and the worker moves on; the task
                                    that uses the result next -- Plot resp. Gather -- ends it.  Non-zero: the task waits for its
                                    own paths, like a reference worker that traces them itself (slower with few workers: the
                                    device only ever has `concurrency` batches to work on) */
    const int* devices;          /* n_devices device indices, rank 0 first (gather, tonemap and output live there);
                                    NULL = device, device + 1, ...  A device may be listed more than once. */
    uint32_t threads;            /* host worker threads; 0 = `concurrency`, as in the reference (app.rs:55,66: one thread per unit
                                    of scheduler depth).  The scheduler's pools are sized by `concurrency` (3 x trace units,
                                    concurrency / 2 plot units, task_scheduler.rs:95-96) -- how many batches the DEVICE can have in
                                    flight; a task here only begins or enqueues device work, so a host with few cores per GPU
                                    sets e.g. concurrency = 16, threads = 2: the pool stays deep, two threads issue it */
} RlAppConfig;

typedef struct RlAppStats {
    uint64_t batches, paths, segments;
    uint64_t tasks[5];           /* executed tasks by RlTaskKind */
    double seconds;
    double kernel_ms;            /* device time in trace kernels */
    float batches_per_sec_mean, batches_per_sec_stddev; /* task_scheduler.rs:308-325 */
    uint32_t tonemaps;
    uint64_t next_batch;         /* first_batch of a run that continues this one */
} RlAppStats;

/* App::new + the worker loops (app.rs:54-111) until max_batches trace tasks are done, gathered and
 * tonemapped once more.  The batches cover the path indices [first_batch * photons_per_batch,
 * (first_batch + max_batches) * photons_per_batch) of every rank's RNG stream exactly once (un-fused: trace task number k, in scheduler order, renders batch k; fused: each plot task renders the
 * batches of its trace units as one launch over the next contiguous range), so the final image does not
 * depend on which worker or unit ran which task.  rgb_out (may be NULL) receives the last RGB8 image. */
int rl_app_run(const RlAppConfig* config, RlAppStats* stats, uint8_t* rgb_out);

#ifdef __cplusplus
}
#endif
#endif /* ROBIGO_LUCULENTA_H */
