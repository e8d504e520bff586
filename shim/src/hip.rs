//! FFI binding of `include/recgraph_hip.h` (the C ABI of the MI355X library) and thin RAII wrappers.
//! UNCOMPILED: no Rust toolchain exists in the image this was written in.
#![allow(non_camel_case_types, dead_code)]

use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_float, c_int};
use std::ptr;

pub const RG_OK: i32 = 0;
pub const RG_READ_BAND_WARNING: u32 = 1;
pub const RG_READ_BAND_NOT_ENOUGH: u32 = 2;
pub const RG_READ_WOULD_PANIC: u32 = 4;
pub const RG_READ_BAD_BASE: u32 = 8;

pub const RG_MODE_GLOBAL_POA: i32 = 0;
pub const RG_MODE_GLOBAL_POA_SCALAR: i32 = 10;
pub const RG_MODE_GAP_POA: i32 = 2;
pub const RG_MODE_LOCAL_POA: i32 = 1;
pub const RG_MODE_GAP_LOCAL_POA: i32 = 3;
pub const RG_SCORE_MISSING: i32 = -536870912;

/// `rg_params` of recgraph_hip.h (field order and types must match the header exactly).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rg_params {
    pub mode: i32,
    pub scores: [i32; 36],
    pub gap_open: i32,
    pub gap_ext: i32,
    pub band_b: c_float,
    pub band_f: c_float,
    pub bta_override: i64,
    pub base_rec_cost: i32,
    pub multi_rec_cost: c_float,
    pub rec_band_width: c_float,
    pub amb_mode: i32,
}

/// `rg_gaf_fields` of recgraph_hip.h.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rg_gaf_fields {
    pub has_record: i32,
    pub empty: i32,
    pub warning: u32,
    pub strand: c_char,
    pub query_length: u64,
    pub query_start: u64,
    pub query_end: u64,
    pub path_length: u64,
    pub path_start: u64,
    pub path_end: u64,
    pub residue_matches_number: u64,
    pub n_path_ids: i64,
    pub comments_len: i64,
}

#[repr(C)]
pub struct rg_graph {
    _private: [u8; 0],
}
#[repr(C)]
pub struct rg_batch {
    _private: [u8; 0],
}

extern "C" {
    pub fn rg_params_default(p: *mut rg_params, mode: i32);
    pub fn rg_graph_create_lnz(
        lnz: *const c_char,
        l: i64,
        pred_off: *const i64,
        pred_rows: *const i64,
        node_id: *const u64,
        out: *mut *mut rg_graph,
    ) -> i32;
    pub fn rg_graph_destroy(g: *mut rg_graph);
    pub fn rg_align_batch(
        g: *const rg_graph,
        p: *const rg_params,
        reads: *const c_char,
        read_off: *const i64,
        nreads: i64,
        out: *mut *mut rg_batch,
    ) -> i32;
    pub fn rg_batch_destroy(b: *mut rg_batch);
    pub fn rg_result_status(b: *const rg_batch, i: i64) -> u32;
    pub fn rg_result_score(b: *const rg_batch, i: i64) -> i32;
    pub fn rg_result_fields(
        b: *const rg_batch,
        i: i64,
        out: *mut rg_gaf_fields,
        path_ids: *mut u64,
        path_cap: i64,
        comments: *mut c_char,
        comments_cap: i64,
    ) -> i32;
    pub fn rg_last_error() -> *const c_char;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rg_last_error()).to_string_lossy().into_owned() }
}

/// Flattened LnzGraph resident on the GPU (`rg_graph`).
pub struct Graph {
    raw: *mut rg_graph,
}

impl Graph {
    /// `lnz` = `'$'` + bases + `'F'`; predecessors of row `i` are `pred_rows[pred_off[i]..pred_off[i+1]]`
    /// (pred_hash order); `node_id[i]` = segment id of row `i` (0 for rows 0 and L-1).
    pub fn from_lnz(lnz: &[u8], pred_off: &[i64], pred_rows: &[i64], node_id: &[u64]) -> Result<Graph, String> {
        let mut raw: *mut rg_graph = ptr::null_mut();
        let rc = unsafe {
            rg_graph_create_lnz(
                lnz.as_ptr() as *const c_char,
                lnz.len() as i64,
                pred_off.as_ptr(),
                pred_rows.as_ptr(),
                node_id.as_ptr(),
                &mut raw,
            )
        };
        if rc != RG_OK {
            return Err(last_error());
        }
        Ok(Graph { raw })
    }
}

impl Drop for Graph {
    fn drop(&mut self) {
        unsafe { rg_graph_destroy(self.raw) }
    }
}

/// One structured alignment record (the fields of `GAFStruct`, gaf_output.rs:6-20).
pub struct Record {
    pub status: u32,
    pub score: i32,
    pub fields: rg_gaf_fields,
    pub path: Vec<usize>,
    pub comments: String,
}

/// create + run + fetch of one read set (`rg_align_batch`); results are read through `record`.
pub struct Batch {
    raw: *mut rg_batch,
}

impl Batch {
    pub fn align(graph: &Graph, params: &rg_params, reads: &[&str]) -> Result<Batch, String> {
        let mut blob = String::new();
        let mut off: Vec<i64> = vec![0];
        for r in reads {
            blob.push_str(r);
            off.push(blob.len() as i64);
        }
        let c = CString::new(blob).map_err(|e| e.to_string())?;
        let mut raw: *mut rg_batch = ptr::null_mut();
        let rc = unsafe { rg_align_batch(graph.raw, params, c.as_ptr(), off.as_ptr(), reads.len() as i64, &mut raw) };
        if rc != RG_OK {
            return Err(last_error());
        }
        Ok(Batch { raw })
    }

    pub fn record(&self, i: usize) -> Result<Record, String> {
        let mut f = rg_gaf_fields::default();
        let rc = unsafe { rg_result_fields(self.raw, i as i64, &mut f, ptr::null_mut(), 0, ptr::null_mut(), 0) };
        if rc != RG_OK {
            return Err(last_error());
        }
        let mut ids: Vec<u64> = vec![0; f.n_path_ids.max(1) as usize];
        let mut com: Vec<u8> = vec![0; f.comments_len as usize + 1];
        let rc = unsafe {
            rg_result_fields(
                self.raw,
                i as i64,
                &mut f,
                ids.as_mut_ptr(),
                ids.len() as i64,
                com.as_mut_ptr() as *mut c_char,
                com.len() as i64,
            )
        };
        if rc != RG_OK {
            return Err(last_error());
        }
        ids.truncate(f.n_path_ids as usize);
        com.truncate(f.comments_len as usize);
        Ok(Record {
            status: unsafe { rg_result_status(self.raw, i as i64) },
            score: unsafe { rg_result_score(self.raw, i as i64) },
            fields: f,
            path: ids.iter().map(|x| *x as usize).collect(),
            comments: String::from_utf8_lossy(&com).into_owned(),
        })
    }
}

impl Drop for Batch {
    fn drop(&mut self) {
        unsafe { rg_batch_destroy(self.raw) }
    }
}

pub fn default_params(mode: i32) -> rg_params {
    let mut p = std::mem::MaybeUninit::<rg_params>::uninit();
    unsafe {
        rg_params_default(p.as_mut_ptr(), mode);
        p.assume_init()
    }
}

// keeps c_int in the import list meaningful for callers that extend the binding
#[allow(dead_code)]
type _Unused = c_int;
