// ffi.rs -- Rust declarations of include/robigo_luculenta.h (the subset App::execute_*_task needs).
// Documentation artefact: this image has no rustc, so the file is compiled only where cargo exists;
// INTEGRATION.md explains how the reference's units wrap these handles.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int};

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

pub enum RlScene {} pub enum RlTraceUnit {} pub enum RlPlotUnit {} pub enum RlGatherUnit {} pub enum RlTonemapUnit {}

extern "C" {
    pub fn rl_last_error() -> *const c_char;
    pub fn rl_device_count() -> c_int;
    pub fn rl_scene_builtin_desc(which: c_int, param: c_int, objects: *mut RlObjectDesc, cap: u32,
                                 n_objects: *mut u32, camera: *mut RlCameraDesc) -> c_int;
    pub fn rl_scene_create(desc: *const RlSceneDesc, device: c_int, out: *mut *mut RlScene) -> c_int;
    pub fn rl_scene_destroy(scene: *mut RlScene) -> c_int;

    pub fn rl_trace_unit_create(device: c_int, id: u32, w: u32, h: u32, n_photons: u32, out: *mut *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_destroy(u: *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_render(u: *mut RlTraceUnit, scene: *const RlScene, seed: u64, stream: u32, first_path: u64) -> c_int;
    pub fn rl_trace_unit_render_fused(u: *mut RlTraceUnit, scene: *const RlScene, plot: *mut RlPlotUnit,
                                      seed: u64, stream: u32, first_path: u64, n_paths: u64) -> c_int;
    pub fn rl_trace_unit_sync(u: *mut RlTraceUnit) -> c_int;
    pub fn rl_trace_unit_photons(u: *mut RlTraceUnit, out: *mut RlMappedPhoton) -> c_int;

    pub fn rl_plot_unit_create(device: c_int, id: u32, w: u32, h: u32, external_xyz: *mut f32, out: *mut *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_destroy(u: *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_plot(u: *mut RlPlotUnit, trace_units: *const *mut RlTraceUnit, n: u32) -> c_int;
    pub fn rl_plot_unit_clear(u: *mut RlPlotUnit) -> c_int;
    pub fn rl_plot_unit_download(u: *mut RlPlotUnit, out: *mut RlVector3) -> c_int;

    pub fn rl_gather_unit_create(device: c_int, w: u32, h: u32, out: *mut *mut RlGatherUnit) -> c_int;
    pub fn rl_gather_unit_destroy(u: *mut RlGatherUnit) -> c_int;
    pub fn rl_gather_unit_accumulate(u: *mut RlGatherUnit, plot: *mut RlPlotUnit) -> c_int;
    pub fn rl_gather_unit_save(u: *mut RlGatherUnit, path: *const c_char) -> c_int;
    pub fn rl_gather_unit_load(u: *mut RlGatherUnit, path: *const c_char) -> c_int;

    pub fn rl_tonemap_unit_create(device: c_int, w: u32, h: u32, out: *mut *mut RlTonemapUnit) -> c_int;
    pub fn rl_tonemap_unit_destroy(u: *mut RlTonemapUnit) -> c_int;
    pub fn rl_tonemap_unit_tonemap(u: *mut RlTonemapUnit, gather: *mut RlGatherUnit) -> c_int;
    pub fn rl_tonemap_unit_rgb(u: *mut RlTonemapUnit, out: *mut u8) -> c_int;
}

pub fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(rl_last_error()) }.to_string_lossy().into_owned();
        panic!("robigo_luculenta: {} ({})", msg, rc);   // the reference panics on every error (app.rs:107,163)
    }
}
