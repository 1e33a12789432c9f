// ffi.rs -- Rust declarations of every entry point of include/robigo_luculenta.h
// (tests/test_abi.py parses both files and compares every function's arity, argument and return types through a
// C -> Rust type map, and every #[repr(C)] struct's field names, order, types and size with the header and with
// the ctypes mirror, so a u32 swapped for a u64 fails the test).
// Documentation artefact: this image has no rustc, so the file is compiled only where cargo exists;
// INTEGRATION.md explains how the reference's units wrap these handles.
#![allow(dead_code)]
use std::os::raw::{c_char, c_double, c_float, c_int};

#[repr(C)] #[derive(Copy, Clone)] pub struct RlVector3 { pub x: f32, pub y: f32, pub z: f32 }          // vector3.rs:20-25
#[repr(C)] #[derive(Copy, Clone)] pub struct RlMappedPhoton { pub x: f32, pub y: f32, pub probability: f32, pub wavelength: f32 } // trace_unit.rs:23-37
#[repr(C)] #[derive(Copy, Clone)] pub struct RlObjectDesc {
    pub surface_kind: u32, pub material_kind: u32, pub v0: RlVector3, pub v1: RlVector3,
    pub f0: f32, pub f1: f32, pub f2: f32, pub f3: f32, pub m0: f32, pub m1: f32, pub m2: f32,
}
#[repr(C)] #[derive(Copy, Clone)] pub struct RlCameraDesc {
    pub phi0: f32, pub phi1: f32, pub alpha0: f32, pub alpha1: f32, pub dist0: f32, pub dist1: f32,
    pub fov_over_pi: f32, pub focal_factor: f32, pub depth_of_field: f32, pub chromatic_abberation: f32,
}
#[repr(C)] pub struct RlSceneDesc { pub n_objects: u32, pub objects: *const RlObjectDesc, pub camera: RlCameraDesc }

pub const RL_TASK_MAX_UNITS: usize = 256;
pub const RL_COMM_ID_BYTES: usize = 128;
#[repr(C)] #[derive(Copy, Clone)] pub struct RlTask {            // enum Task by value, task_scheduler.rs:26-41
    pub kind: u32,      // 0 Sleep, 1 Trace, 2 Plot, 3 Gather, 4 Tonemap
    pub unit: u32, pub n_units: u32, pub units: [u32; RL_TASK_MAX_UNITS],
}
#[repr(C)] pub struct RlAppConfig {
    pub width: u32, pub height: u32, pub device: c_int, pub concurrency: u32, pub photons_per_batch: u32,
    pub seed: u64, pub stream: u32, pub builtin_scene: c_int, pub builtin_param: c_int, pub max_batches: u64,
    pub tonemap_interval_ms: i64, pub fused: c_int, pub output_ppm: *const c_char, pub checkpoint: *const c_char,
    pub resume: c_int, pub verbose: c_int, pub sleep_us: u32, pub first_batch: u64, pub n_devices: u32, pub blocking_trace: c_int, pub devices: *const c_int, pub threads: u32,
}
#[repr(C)] #[derive(Copy, Clone, Default)] pub struct RlAppStats {
    pub batches: u64, pub paths: u64, pub segments: u64, pub tasks: [u64; 5], pub seconds: c_double, pub kernel_ms: c_double,
    pub batches_per_sec_mean: c_float, pub batches_per_sec_stddev: c_float, pub tonemaps: u32, pub next_batch: u64,
}

pub enum RlScene {} pub enum RlTraceUnit {} pub enum RlPlotUnit {} pub enum RlGatherUnit {} pub enum RlTonemapUnit {}
pub enum RlScheduler {} pub enum RlComm {}

extern "C" {
    pub fn rl_last_error() -> *const c_char;
    pub fn rl_device_count() -> c_int;
    pub fn rl_device_pci_bus_id(device: c_int, out: *mut c_char, cap: u32) -> c_int;
    pub fn rl_version() -> *const c_char;
    pub fn rl_build_id() -> *const c_char;
    pub fn rl_scene_builtin_desc(which: c_int, param: c_int, objects: *mut RlObjectDesc, cap: u32,
                                 n_objects: *mut u32, camera: *mut RlCameraDesc) -> c_int;
    pub fn rl_scene_desc_save(path: *const c_char, desc: *const RlSceneDesc) -> c_int;
    pub fn rl_scene_desc_load(path: *const c_char, objects: *mut RlObjectDesc, cap: u32, n_objects: *mut u32,
                              camera: *mut RlCameraDesc) -> c_int;
    pub fn rl_scene_create(desc: *const RlSceneDesc, device: c_int, out: *mut *mut RlScene) -> c_int;
    pub fn rl_scene_destroy(scene: *mut RlScene) -> c_int;

    pub fn rl_trace_unit_create(device: c_int, id: u32, w: u32, h: u32, n_photons: u32, out: *mut *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_destroy(u: *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_set_fetch(u: *mut RlTraceUnit, primitive_fetch: c_int) -> c_int;   // 0 LDS, 1 global
    pub fn rl_trace_unit_render(u: *mut RlTraceUnit, scene: *const RlScene, seed: u64, stream: u32, first_path: u64) -> c_int;
    pub fn rl_trace_unit_render_begin(u: *mut RlTraceUnit, scene: *const RlScene, seed: u64, stream: u32, first_path: u64) -> c_int;
    pub fn rl_trace_unit_render_end(u: *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_render_async(u: *mut RlTraceUnit, scene: *const RlScene, seed: u64, stream: u32, first_path: u64) -> c_int;
    pub fn rl_trace_unit_render_fused(u: *mut RlTraceUnit, scene: *const RlScene, plot: *mut RlPlotUnit,
                                      seed: u64, stream: u32, first_path: u64, n_paths: u64) -> c_int;
    pub fn rl_trace_unit_render_fused_sync(u: *mut RlTraceUnit, scene: *const RlScene, plot: *mut RlPlotUnit,
                                           seed: u64, stream: u32, first_path: u64, n_paths: u64) -> c_int;
    pub fn rl_trace_unit_render_fused_begin(u: *mut RlTraceUnit, scene: *const RlScene, plot: *mut RlPlotUnit,
                                            seed: u64, stream: u32, first_path: u64, n_paths: u64) -> c_int;
    pub fn rl_trace_unit_sync(u: *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_photons(u: *mut RlTraceUnit, out: *mut RlMappedPhoton) -> c_int;
    pub fn rl_trace_unit_stats(u: *mut RlTraceUnit, paths: *mut u64, segments: *mut u64, kernel_ms: *mut c_double) -> c_int;

    pub fn rl_plot_unit_create(device: c_int, id: u32, w: u32, h: u32, external_xyz: *mut f32, out: *mut *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_destroy(u: *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_plot(u: *mut RlPlotUnit, trace_units: *const *mut RlTraceUnit, n: u32) -> c_int;
    pub fn rl_plot_unit_clear(u: *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_sync(u: *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_device_buffer(u: *mut RlPlotUnit, device_xyz: *mut *mut f32) -> c_int;
    pub fn rl_plot_unit_download(u: *mut RlPlotUnit, out: *mut RlVector3) -> c_int;
    pub fn rl_plot_unit_upload(u: *mut RlPlotUnit, input: *const RlVector3) -> c_int;

    pub fn rl_gather_unit_create(device: c_int, w: u32, h: u32, out: *mut *mut RlGatherUnit) -> c_int;
    pub fn rl_gather_unit_destroy(u: *mut RlGatherUnit) -> c_int;
    pub fn rl_gather_unit_accumulate(u: *mut RlGatherUnit, plot: *mut RlPlotUnit) -> c_int;
    pub fn rl_gather_unit_save(u: *mut RlGatherUnit, path: *const c_char) -> c_int;
    pub fn rl_gather_unit_load(u: *mut RlGatherUnit, path: *const c_char) -> c_int;
    pub fn rl_gather_unit_download(u: *mut RlGatherUnit, tristimulus: *mut RlVector3, compensation: *mut RlVector3) -> c_int;
    pub fn rl_gather_unit_sync(u: *mut RlGatherUnit) -> c_int;

    // The GatherUnit-time exchange between GPUs (RCCL over xGMI, bound at run time).
    pub fn rl_comm_unique_id(id: *mut u8) -> c_int;                                  // RL_COMM_ID_BYTES bytes
    pub fn rl_comm_init_rank(id: *const u8, world: c_int, rank: c_int, device: c_int, out: *mut *mut RlComm) -> c_int;
    pub fn rl_comm_init_all(devices: *const c_int, n: c_int, out: *mut *mut RlComm) -> c_int;
    pub fn rl_comm_destroy(comm: *mut RlComm) -> c_int;
    pub fn rl_comm_rank(comm: *const RlComm, rank: *mut c_int, world: *mut c_int) -> c_int;
    pub fn rl_comm_info(comm: *const RlComm, rank: *mut c_int, world: *mut c_int, rccl_version: *mut c_int, library_path: *mut c_char, path_cap: u32) -> c_int;
    pub fn rl_comm_group_start() -> c_int;
    pub fn rl_comm_group_end() -> c_int;
    pub fn rl_plot_unit_reduce(u: *mut RlPlotUnit, comm: *mut RlComm, root: c_int) -> c_int;
    pub fn rl_plot_unit_exchange_stats(u: *mut RlPlotUnit, exchanges: *mut u64, device_ms: *mut f64) -> c_int;
    pub fn rl_plot_unit_add(dst: *mut RlPlotUnit, src: *mut RlPlotUnit) -> c_int;
    pub fn rl_gather_unit_allreduce(gather: *mut RlGatherUnit, plot: *mut RlPlotUnit, comm: *mut RlComm) -> c_int;

    pub fn rl_tonemap_unit_create(device: c_int, w: u32, h: u32, out: *mut *mut RlTonemapUnit) -> c_int;
    pub fn rl_tonemap_unit_destroy(u: *mut RlTonemapUnit) -> c_int;
    pub fn rl_tonemap_unit_tonemap(u: *mut RlTonemapUnit, gather: *mut RlGatherUnit) -> c_int;
    pub fn rl_tonemap_unit_rgb(u: *mut RlTonemapUnit, out: *mut u8) -> c_int;
    pub fn rl_tonemap_unit_srgb_float(u: *mut RlTonemapUnit, out: *mut f32, max_intensity: *mut f32) -> c_int;

    // TaskScheduler over unit ids (task_scheduler.rs:91-182, 308-325) and the whole App (app.rs:48-164), for
    // hosts that do not keep the reference's own scheduler.
    pub fn rl_scheduler_create(concurrency: u32, tonemap_interval_ms: i64, out: *mut *mut RlScheduler) -> c_int;
    pub fn rl_scheduler_destroy(s: *mut RlScheduler) -> c_int;
    pub fn rl_scheduler_get_new_task(s: *mut RlScheduler, completed: *const RlTask, now_ms: i64, next: *mut RlTask) -> c_int;
    pub fn rl_scheduler_performance(s: *mut RlScheduler, mean: *mut f32, stddev: *mut f32) -> c_int;
    pub fn rl_app_run(config: *const RlAppConfig, stats: *mut RlAppStats, rgb_out: *mut u8) -> c_int;

}

pub fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(rl_last_error()) }.to_string_lossy().into_owned();
        panic!("robigo_luculenta: {} ({})", msg, rc);   // the reference panics on every error (app.rs:107,163)
    }
}
