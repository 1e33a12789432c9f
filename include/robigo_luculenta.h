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

const char* rl_last_error(void);
/* Number of visible HIP devices (0 when there is none; never fails). */
int rl_device_count(void);
const char* rl_version(void);

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
 * Complete (device-synchronised) on return. */
int rl_trace_unit_render(RlTraceUnit* unit, const RlScene* scene, uint64_t seed, uint32_t stream,
                         uint64_t first_path_index);
/* Fused TraceUnit::render + PlotUnit::plot (trace_unit.rs:151-168 + plot_unit.rs:87-95): traces
 * n_paths paths (any count, not limited to the unit's batch size) and splats every non-zero
 * contribution straight into `plot` with f32 atomics; mapped_photons is not written.  After this a
 * rl_plot_unit_plot for the same photons must NOT be issued.  Asynchronous on the unit's stream;
 * rl_trace_unit_sync() or any download waits for it. */
int rl_trace_unit_render_fused(RlTraceUnit* unit, const RlScene* scene, RlPlotUnit* plot, uint64_t seed,
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
/* PlotUnit::plot(&mut self, &[MappedPhoton]) for each given trace unit (app.rs:136-141). */
int rl_plot_unit_plot(RlPlotUnit* unit, RlTraceUnit* const* trace_units, uint32_t n_trace_units);
/* PlotUnit::clear (plot_unit.rs:98-102). */
int rl_plot_unit_clear(RlPlotUnit* unit);
/* Device pointer of tristimulus_buffer (plot_unit.rs:35): width*height RlVector3, row-major. */
int rl_plot_unit_device_buffer(RlPlotUnit* unit, float** device_xyz);
int rl_plot_unit_download(RlPlotUnit* unit, RlVector3* out);

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

#define RL_TASK_MAX_UNITS 64

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
    uint32_t concurrency;        /* worker threads; the reference uses num_cpus::get() (app.rs:55) */
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
} RlAppConfig;

typedef struct RlAppStats {
    uint64_t batches, paths, segments;
    uint64_t tasks[5];           /* executed tasks by RlTaskKind */
    double seconds;
    double kernel_ms;            /* device time in trace kernels */
    float batches_per_sec_mean, batches_per_sec_stddev; /* task_scheduler.rs:308-325 */
    uint32_t tonemaps;
} RlAppStats;

/* App::new + the worker loops (app.rs:54-111) until max_batches trace tasks are done, gathered and
 * tonemapped once more.  The batches cover the path indices [0, max_batches * photons_per_batch) exactly
 * once (un-fused: trace task number k, in scheduler order, renders batch k; fused: each plot task renders the
 * batches of its trace units as one launch over the next contiguous range), so the final image does not
 * depend on which worker or unit ran which task.  rgb_out (may be NULL) receives the last RGB8 image. */
int rl_app_run(const RlAppConfig* config, RlAppStats* stats, uint8_t* rgb_out);

/* ---- diagnostics -------------------------------------------------------------------------------- */

/* Not a reference interface: evaluates the shared numerics header (csrc/rl_math.h) on the GPU so a
 * test can check that the hipcc and g++ builds agree bit-for-bit.  fn: 0 sin, 1 cos, 2 tan, 3 exp,
 * 4 ln, 5 acos, 6 SF10 index of refraction (material.rs:203-213), 7 sqrt, 8 x[i] / x[i+1 mod n],
 * 9 x^(1/2.4) (srgb.rs:24), 10 the Russian-roulette decision (trace_unit.rs:122-125) for the triples
 * (x[i], x[m+i], x[2m+i]) = (rand, continue_chance, intensity), i < m = n / 3, result 1 or 0 in y[i]. */
int rl_debug_math_probe(int device, int fn, const float* x, float* y, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* ROBIGO_LUCULENTA_H */
