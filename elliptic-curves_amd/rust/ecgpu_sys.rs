//! ecgpu_sys.rs — raw FFI declarations of libecgpu.so, GENERATED from include/ecgpu.h by tools/gen_rust_sys.py.
//! Do not edit: regenerate.  tests/test_abi.py checks this file against the header (name, arity, types).
//! The safe adapters behind the reference's traits are in ecgpu_shim.rs.
#![allow(non_camel_case_types, dead_code)]

use core::ffi::{c_char, c_int, c_void};

/// `ecgpu_ctx`: one GPU, its stream, its device-resident tables (opaque).
#[repr(C)]
pub struct EcgpuCtx {
    _private: [u8; 0],
}
/// `ecgpu_group`: one context per GPU of a node, driven from one process (opaque).
#[repr(C)]
pub struct EcgpuGroup {
    _private: [u8; 0],
}

pub const ECGPU_K256: c_int = 0;
pub const ECGPU_P256: c_int = 1;
pub const ECGPU_P384: c_int = 2;
pub const ECGPU_SM2: c_int = 3;
pub const ECGPU_P224: c_int = 4;
pub const ECGPU_P192: c_int = 5;
pub const ECGPU_P521: c_int = 6;
pub const ECGPU_BP256: c_int = 7;
pub const ECGPU_BP384: c_int = 8;
pub const ECGPU_BP256T1: c_int = 9;
pub const ECGPU_BP384T1: c_int = 10;
pub const ECGPU_BIGN256: c_int = 11;
pub const ECGPU_OK: c_int = 0;
pub const ECGPU_ERR_CURVE: c_int = -1;
pub const ECGPU_ERR_SCALAR_RANGE: c_int = -2;
pub const ECGPU_ERR_POINT: c_int = -3;
pub const ECGPU_ERR_NO_DEVICE: c_int = -4;
pub const ECGPU_ERR_HIP: c_int = -5;
pub const ECGPU_ERR_OOM: c_int = -6;
pub const ECGPU_ERR_ARG: c_int = -7;
pub const ECGPU_TABLE_ADAPTIVE: c_int = 0;
pub const ECGPU_TABLE_EAGER: c_int = 1;
pub const ECGPU_EXCHANGE_PEER: c_int = 1;
pub const ECGPU_EXCHANGE_RCCL: c_int = 2;

#[link(name = "ecgpu")]
unsafe extern "C" {
    pub fn ecgpu_device_count() -> c_int;
    pub fn ecgpu_init(ctx: *mut *mut EcgpuCtx, device: c_int) -> c_int;
    pub fn ecgpu_destroy(ctx: *mut EcgpuCtx);
    pub fn ecgpu_last_error(ctx: *const EcgpuCtx) -> *const c_char;
    pub fn ecgpu_field_bytes(curve: c_int) -> usize;
    pub fn ecgpu_set_stream(ctx: *mut EcgpuCtx, stream: *mut c_void) -> c_int;
    pub fn ecgpu_host_alloc(ctx: *mut EcgpuCtx, bytes: usize) -> *mut c_void;
    pub fn ecgpu_host_free(ctx: *mut EcgpuCtx, p: *mut c_void);
    pub fn ecgpu_dev_alloc(ctx: *mut EcgpuCtx, bytes: usize) -> *mut c_void;
    pub fn ecgpu_dev_free(ctx: *mut EcgpuCtx, d_ptr: *mut c_void);
    pub fn ecgpu_copy_to_device(ctx: *mut EcgpuCtx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn ecgpu_copy_to_host(ctx: *mut EcgpuCtx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn ecgpu_set_table_policy(ctx: *mut EcgpuCtx, policy: c_int) -> c_int;
    pub fn ecgpu_set_table_budget(ctx: *mut EcgpuCtx, max_table_bytes: usize) -> c_int;
    pub fn ecgpu_base_table_info(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        window_bits: *mut c_int,
        table_bytes: *mut usize,
        build_ms: *mut f64,
    ) -> c_int;
    pub fn ecgpu_set_base_window(ctx: *mut EcgpuCtx, curve: c_int, window_bits: c_int) -> c_int;
    pub fn ecgpu_set_msm_window(ctx: *mut EcgpuCtx, window_bits: c_int) -> c_int;
    pub fn ecgpu_set_async(ctx: *mut EcgpuCtx, on: c_int) -> c_int;
    pub fn ecgpu_synchronize(ctx: *mut EcgpuCtx) -> c_int;
    pub fn ecgpu_set_msm_lanes(ctx: *mut EcgpuCtx, lanes: c_int) -> c_int;
    pub fn ecgpu_wipe(ctx: *mut EcgpuCtx) -> c_int;
    pub fn ecgpu_batch_mul_base(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_compressed(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        n: usize,
        out_x: *mut u8,
        out_tag: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_msm(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_and_mul_add(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        a_scalars: *const u8,
        b_scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_normalize(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        points_xyz: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_compressed_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        n: usize,
        d_out_x: *mut c_void,
        d_out_tag: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_mul_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_msm_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_and_mul_add_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_a_scalars: *const c_void,
        d_b_scalars: *const c_void,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_normalize_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_points_xyz: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_point_sum(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_point_sum_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_msm_parts_bytes(ctx: *mut EcgpuCtx, curve: c_int, plan_terms: usize) -> usize;
    pub fn ecgpu_msm_plan_window(ctx: *mut EcgpuCtx, curve: c_int, plan_terms: usize) -> c_int;
    pub fn ecgpu_msm_parts_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        plan_terms: usize,
        d_parts: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_msm_parts_join_dev(ctx: *mut EcgpuCtx, d_parts: *const c_void) -> c_int;
    pub fn ecgpu_msm_finish_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_parts_all: *const c_void,
        nranks: c_int,
        plan_terms: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_group_init(group: *mut *mut EcgpuGroup, devices: *const c_int, ndev: c_int) -> c_int;
    pub fn ecgpu_group_destroy(group: *mut EcgpuGroup);
    pub fn ecgpu_group_size(group: *const EcgpuGroup) -> c_int;
    pub fn ecgpu_group_ctx(group: *mut EcgpuGroup, i: c_int) -> *mut EcgpuCtx;
    pub fn ecgpu_group_last_error(group: *const EcgpuGroup) -> *const c_char;
    pub fn ecgpu_group_exchange(group: *const EcgpuGroup) -> *const c_char;
    pub fn ecgpu_group_exchange_reason(group: *const EcgpuGroup) -> *const c_char;
    pub fn ecgpu_group_set_exchange(group: *mut EcgpuGroup, mode: c_int) -> c_int;
    pub fn ecgpu_group_set_msm_window(group: *mut EcgpuGroup, window_bits: c_int) -> c_int;
    pub fn ecgpu_group_set_exchange_timeout(group: *mut EcgpuGroup, seconds: f64) -> c_int;
    pub fn ecgpu_group_msm(
        group: *mut EcgpuGroup,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_group_msm_dev(
        group: *mut EcgpuGroup,
        curve: c_int,
        d_scalars: *const *const c_void,
        d_points_xy: *const *const c_void,
        d_points_inf: *const *const c_void,
        n_per_device: *const usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_group_batch_mul_base(
        group: *mut EcgpuGroup,
        curve: c_int,
        scalars: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_group_batch_mul(
        group: *mut EcgpuGroup,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_group_ecdsa_verify_batch(
        group: *mut EcgpuGroup,
        curve: c_int,
        z: *const u8,
        r: *const u8,
        s: *const u8,
        q_xy: *const u8,
        n: usize,
        reject_high_s: c_int,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_group_ecdsa_verify_msg_batch(
        group: *mut EcgpuGroup,
        curve: c_int,
        q_xy: *const u8,
        msgs: *const u8,
        msg_len: usize,
        sigs: *const u8,
        n: usize,
        reject_high_s: c_int,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_group_ecdsa_recover_batch(
        group: *mut EcgpuGroup,
        curve: c_int,
        z: *const u8,
        r: *const u8,
        s: *const u8,
        recid: *const u8,
        n: usize,
        reject_high_s: c_int,
        out_xy: *mut u8,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_k256_glv_decompose(
        ctx: *mut EcgpuCtx,
        scalars: *const u8,
        n: usize,
        r1: *mut u8,
        r2: *mut u8,
    ) -> c_int;
    pub fn ecgpu_ecdsa_verify_batch(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        z: *const u8,
        r: *const u8,
        s: *const u8,
        q_xy: *const u8,
        n: usize,
        reject_high_s: c_int,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_ecdsa_verify_batch_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_z: *const c_void,
        d_r: *const c_void,
        d_s: *const c_void,
        d_q_xy: *const c_void,
        n: usize,
        reject_high_s: c_int,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_ecdsa_verify_msg_batch(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        q_xy: *const u8,
        msgs: *const u8,
        msg_len: usize,
        sigs: *const u8,
        n: usize,
        reject_high_s: c_int,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_ecdsa_verify_msg_batch_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_q_xy: *const c_void,
        d_msgs: *const c_void,
        msg_len: usize,
        d_sigs: *const c_void,
        n: usize,
        reject_high_s: c_int,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_ecdsa_recover_batch(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        z: *const u8,
        r: *const u8,
        s: *const u8,
        recid: *const u8,
        n: usize,
        reject_high_s: c_int,
        out_xy: *mut u8,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_ecdsa_recover_batch_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_z: *const c_void,
        d_r: *const c_void,
        d_s: *const c_void,
        d_recid: *const c_void,
        n: usize,
        reject_high_s: c_int,
        d_out_xy: *mut c_void,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_schnorr_verify_batch(
        ctx: *mut EcgpuCtx,
        e: *const u8,
        r: *const u8,
        s: *const u8,
        p_xy: *const u8,
        n: usize,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_schnorr_verify_batch_dev(
        ctx: *mut EcgpuCtx,
        d_e: *const c_void,
        d_r: *const c_void,
        d_s: *const c_void,
        d_p_xy: *const c_void,
        n: usize,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_sm2dsa_verify_batch(
        ctx: *mut EcgpuCtx,
        e: *const u8,
        r: *const u8,
        s: *const u8,
        q_xy: *const u8,
        n: usize,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_sm2dsa_verify_batch_dev(
        ctx: *mut EcgpuCtx,
        d_e: *const c_void,
        d_r: *const c_void,
        d_s: *const c_void,
        d_q_xy: *const c_void,
        n: usize,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_sm2dsa_verify_msg_batch(
        ctx: *mut EcgpuCtx,
        distid: *const u8,
        distid_len: usize,
        q_xy: *const u8,
        msgs: *const u8,
        msg_len: usize,
        sigs: *const u8,
        n: usize,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_sm2dsa_verify_msg_batch_dev(
        ctx: *mut EcgpuCtx,
        d_distid: *const c_void,
        distid_len: usize,
        d_q_xy: *const c_void,
        d_msgs: *const c_void,
        msg_len: usize,
        d_sigs: *const c_void,
        n: usize,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_bign_verify_batch(
        ctx: *mut EcgpuCtx,
        h: *const u8,
        sigs: *const u8,
        q_xy: *const u8,
        n: usize,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_bign_verify_batch_dev(
        ctx: *mut EcgpuCtx,
        d_h: *const c_void,
        d_sigs: *const c_void,
        d_q_xy: *const c_void,
        n: usize,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_bign_verify_msg_batch(
        ctx: *mut EcgpuCtx,
        q_xy: *const u8,
        msgs: *const u8,
        msg_len: usize,
        sigs: *const u8,
        n: usize,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_bign_verify_msg_batch_dev(
        ctx: *mut EcgpuCtx,
        d_q_xy: *const c_void,
        d_msgs: *const c_void,
        msg_len: usize,
        d_sigs: *const c_void,
        n: usize,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_schnorr_verify_raw_batch(
        ctx: *mut EcgpuCtx,
        pk_x: *const u8,
        msgs: *const u8,
        msg_len: usize,
        sigs: *const u8,
        n: usize,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_schnorr_verify_raw_batch_dev(
        ctx: *mut EcgpuCtx,
        d_pk_x: *const c_void,
        d_msgs: *const c_void,
        msg_len: usize,
        d_sigs: *const c_void,
        n: usize,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_ecdh(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        n: usize,
        out_x: *mut u8,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_ecdh_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        n: usize,
        d_out_x: *mut c_void,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_ct(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul_base_ct_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_mul_ct(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul_ct_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_ecdh_ct(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        n: usize,
        out_x: *mut u8,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_ecdh_ct_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        n: usize,
        d_out_x: *mut c_void,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_lincomb_ct(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_xy: *const u8,
        points_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_lincomb_ct_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_xy: *const c_void,
        d_points_inf: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_msm_compressed(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_x: *const u8,
        points_tag: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_msm_compressed_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_x: *const c_void,
        d_points_tag: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_mul_compressed(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        scalars: *const u8,
        points_x: *const u8,
        points_tag: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_mul_compressed_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_scalars: *const c_void,
        d_points_x: *const c_void,
        d_points_tag: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_out_inf: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_batch_decompress(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        xs: *const u8,
        y_is_odd: *const u8,
        n: usize,
        out_xy: *mut u8,
        ok: *mut u8,
    ) -> c_int;
    pub fn ecgpu_batch_decompress_dev(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        d_xs: *const c_void,
        d_y_is_odd: *const c_void,
        n: usize,
        d_out_xy: *mut c_void,
        d_ok: *mut c_void,
    ) -> c_int;
    pub fn ecgpu_selftest_field(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        op: c_int,
        a: *const u8,
        b: *const u8,
        n: usize,
        out: *mut u8,
    ) -> c_int;
    pub fn ecgpu_selftest_point(
        ctx: *mut EcgpuCtx,
        curve: c_int,
        op: c_int,
        p_xy: *const u8,
        p_inf: *const u8,
        q_xy: *const u8,
        q_inf: *const u8,
        n: usize,
        out_xy: *mut u8,
        out_inf: *mut u8,
    ) -> c_int;
    pub fn ecgpu_valu_probe(ctx: *mut EcgpuCtx, which: c_int, ops_per_sec: *mut f64) -> c_int;
    pub fn ecgpu_last_timing(ctx: *const EcgpuCtx, name: *const c_char, ms: *mut f64) -> c_int;
    pub fn ecgpu_set_timing(ctx: *mut EcgpuCtx, on: c_int) -> c_int;
    pub fn ecgpu_version() -> *const c_char;
}
