//! Safe RAII wrappers over the C ABI of librecgraph_hip.so.  The raw `extern "C"` binding of EVERY symbol of
//! `include/recgraph_hip.h` lives in `hip_ffi.rs` (generated from the header by tools/gen_rust_ffi.py, re-exported here).
//! UNCOMPILED: no Rust toolchain exists in the image this was written in.
#![allow(dead_code)]

use std::ffi::{CStr, CString};
use std::os::raw::c_char;
use std::ptr;

pub use crate::hip_ffi::*;

pub fn last_error() -> String {
    unsafe { CStr::from_ptr(rg_last_error()).to_string_lossy().into_owned() }
}

fn check(rc: i32) -> Result<(), String> {
    if rc == RG_OK {
        Ok(())
    } else {
        Err(format!("recgraph_hip error {}: {}", rc, last_error()))
    }
}

/// `rg_fasta_check` over a whole file, block by block: the file-level panic of sequences::get_sequences
/// (sequences.rs:41-43, "wrong fasta file format") ahead of any output.  Returns the number of reads.
pub fn fasta_check_file(path: &str) -> Result<i64, String> {
    use std::io::Read;
    let mut file = std::fs::File::open(path).map_err(|e| e.to_string())?;
    let mut block = vec![0u8; 4 << 20];
    let mut state = [0i64; 4];
    let mut n: i64 = 0;
    loop {
        let got = file.read(&mut block).map_err(|e| e.to_string())?;
        check(unsafe { rg_fasta_check(block.as_ptr() as *const c_char, got as i64, (got == 0) as i32, state.as_mut_ptr(), &mut n) })?;
        if got == 0 {
            return Ok(n);
        }
    }
}

pub fn default_params(mode: i32) -> rg_params {
    let mut p = std::mem::MaybeUninit::<rg_params>::uninit();
    unsafe {
        rg_params_default(p.as_mut_ptr(), mode);
        p.assume_init()
    }
}

/// Flattened graph resident on the GPU (`rg_graph`): LnzGraph view, plus the PathGraph view when the GFA has P lines.
pub struct Graph {
    pub(crate) raw: *mut rg_graph,
}
// immutable after creation, shareable across threads and devices (recgraph_hip.h)
unsafe impl Send for Graph {}
unsafe impl Sync for Graph {}

impl Graph {
    /// graph::read_graph / pathwise_graph::read_graph_w_path (graph.rs:11, pathwise_graph.rs:127) on GFA text.
    pub fn from_gfa_text(text: &str) -> Result<Graph, String> {
        let mut raw: *mut rg_graph = ptr::null_mut();
        check(unsafe { rg_graph_from_gfa(text.as_ptr() as *const c_char, text.len() as i64, &mut raw) })?;
        Ok(Graph { raw })
    }

    /// `lnz` = `'$'` + bases + `'F'`; predecessors of row `i` are `pred_rows[pred_off[i]..pred_off[i+1]]`
    /// (pred_hash order); `node_id[i]` = segment id of row `i` (0 for rows 0 and L-1).
    pub fn from_lnz(lnz: &[u8], pred_off: &[i64], pred_rows: &[i64], node_id: &[u64]) -> Result<Graph, String> {
        let mut raw: *mut rg_graph = ptr::null_mut();
        check(unsafe {
            rg_graph_create_lnz(lnz.as_ptr() as *const c_char, lnz.len() as i64, pred_off.as_ptr(), pred_rows.as_ptr(), node_id.as_ptr(), &mut raw)
        })?;
        Ok(Graph { raw })
    }

    pub fn rows(&self) -> i64 {
        unsafe { rg_graph_rows(self.raw) }
    }
    pub fn paths(&self) -> i32 {
        unsafe { rg_graph_paths(self.raw) }
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

fn record_of(raw: *const rg_batch, i: usize) -> Result<Record, String> {
    let mut f = std::mem::MaybeUninit::<rg_gaf_fields>::zeroed();
    check(unsafe { rg_result_fields(raw, i as i64, f.as_mut_ptr(), ptr::null_mut(), 0, ptr::null_mut(), 0) })?;
    let mut f = unsafe { f.assume_init() };
    let mut ids: Vec<u64> = vec![0; f.n_path_ids.max(1) as usize];
    let mut com: Vec<u8> = vec![0; f.comments_len as usize + 1];
    check(unsafe { rg_result_fields(raw, i as i64, &mut f, ids.as_mut_ptr(), ids.len() as i64, com.as_mut_ptr() as *mut c_char, com.len() as i64) })?;
    ids.truncate(f.n_path_ids as usize);
    com.truncate(f.comments_len as usize);
    Ok(Record {
        status: unsafe { rg_result_status(raw, i as i64) },
        score: unsafe { rg_result_score(raw, i as i64) },
        fields: f,
        path: ids.iter().map(|x| *x as usize).collect(),
        comments: String::from_utf8_lossy(&com).into_owned(),
    })
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
        check(unsafe { rg_align_batch(graph.raw, params, c.as_ptr(), off.as_ptr(), reads.len() as i64, &mut raw) })?;
        Ok(Batch { raw })
    }
    pub fn record(&self, i: usize) -> Result<Record, String> {
        record_of(self.raw, i)
    }
}

impl Drop for Batch {
    fn drop(&mut self) {
        unsafe { rg_batch_destroy(self.raw) }
    }
}

/// sequences::get_sequences (sequences.rs:5-45) inside the library: bases blob + offsets + names.
pub struct Reads {
    raw: *mut rg_reads,
}

impl Reads {
    pub fn from_fasta_text(text: &[u8]) -> Result<Reads, String> {
        let mut raw: *mut rg_reads = ptr::null_mut();
        check(unsafe { rg_reads_from_fasta(text.as_ptr() as *const c_char, text.len() as i64, &mut raw) })?;
        Ok(Reads { raw })
    }
    pub fn len(&self) -> usize {
        unsafe { rg_reads_count(self.raw) as usize }
    }
    pub fn name(&self, i: usize) -> String {
        unsafe { CStr::from_ptr(*rg_reads_names(self.raw).add(i)).to_string_lossy().into_owned() }
    }
}

impl Drop for Reads {
    fn drop(&mut self) {
        unsafe { rg_reads_destroy(self.raw) }
    }
}

/// One tile of a stream: the stdout text of its reads (exactly what the reference prints) + per-read status bits.
pub struct Tile {
    pub first_read: usize,
    pub text: Vec<u8>,
    pub text_off: Vec<i64>,
    pub status: Vec<u32>,
    pub score: Vec<i32>,
    pub device: i32,
}

/// `rg_stream`: the read loop of main.rs (:56,174,257,297-312) as the pipeline hidden behind the C ABI — tiles of reads
/// pulled by `handles_per_device` batch handles per device from one queue, results in input order.
pub struct Stream {
    raw: *mut rg_stream,
}
// The C ABI lets one thread push / feed while another takes results (rg_stream_push: "any thread, any time"; one thread
// at a time in rg_stream_next): the feeder thread of main_loop.rs shares the stream by reference.
unsafe impl Send for Stream {}
unsafe impl Sync for Stream {}

impl Stream {
    /// `devices`: None = every visible GPU.
    pub fn new(graph: &Graph, params: &rg_params, devices: Option<&[i32]>, opts: Option<rg_stream_opts>) -> Result<Stream, String> {
        let mut o = std::mem::MaybeUninit::<rg_stream_opts>::uninit();
        let o = match opts {
            Some(x) => x,
            None => unsafe {
                rg_stream_opts_default(o.as_mut_ptr());
                o.assume_init()
            },
        };
        let mut raw: *mut rg_stream = ptr::null_mut();
        let (dp, dn) = match devices {
            Some(d) => (d.as_ptr(), d.len() as i32),
            None => (ptr::null(), 0),
        };
        check(unsafe { rg_stream_create(graph.raw, params, dp, dn, &o, &mut raw) })?;
        Ok(Stream { raw })
    }

    /// FASTA text parsed inside the library, its reads pushed tile by tile while the rest is parsed.
    pub fn push_fasta(&self, text: &[u8]) -> Result<usize, String> {
        let mut n: i64 = 0;
        check(unsafe { rg_stream_push_fasta(self.raw, text.as_ptr() as *const c_char, text.len() as i64, &mut n) })?;
        Ok(n as usize)
    }

    /// The same for a text that arrives in pieces (a file read block by block); `last` closes it.
    pub fn feed_fasta(&self, piece: &[u8], last: bool) -> Result<usize, String> {
        let mut n: i64 = 0;
        check(unsafe { rg_stream_feed_fasta(self.raw, piece.as_ptr() as *const c_char, piece.len() as i64, last as i32, &mut n) })?;
        Ok(n as usize)
    }

    /// Tiles pushed and not yet taken with `next`.
    pub fn pending(&self) -> i64 {
        unsafe { rg_stream_pending(self.raw) }
    }

    /// Options with the library's defaults (`rg_stream_opts_default`), for `Stream::new(.., Some(opts))`.
    pub fn default_opts() -> rg_stream_opts {
        let mut o = std::mem::MaybeUninit::<rg_stream_opts>::uninit();
        unsafe {
            rg_stream_opts_default(o.as_mut_ptr());
            o.assume_init()
        }
    }

    pub fn push(&self, reads: &Reads) -> Result<(), String> {
        check(unsafe { rg_stream_push(self.raw, rg_reads_bases(reads.raw), rg_reads_offsets(reads.raw), rg_reads_count(reads.raw), rg_reads_names(reads.raw)) })
    }

    pub fn finish(&self) -> Result<(), String> {
        check(unsafe { rg_stream_finish(self.raw) })
    }

    /// Error exit: queued tiles are dropped and every thread blocked in `push` / `feed_fasta` / `next` returns `Err`
    /// (`rg_stream_abort`).  Call it before leaving a scope that joins a feeder thread on any error path.
    pub fn abort(&self) {
        unsafe { rg_stream_abort(self.raw) };
    }

    /// The next tile in input order (blocks); Ok(None) after `finish` when everything was delivered.
    pub fn next(&self) -> Result<Option<Tile>, String> {
        let mut r = std::mem::MaybeUninit::<rg_stream_result>::zeroed();
        let rc = unsafe { rg_stream_next(self.raw, r.as_mut_ptr()) };
        if rc == RG_STREAM_END {
            return Ok(None);
        }
        check(rc)?;
        let r = unsafe { r.assume_init() };
        let n = r.nreads as usize;
        unsafe {
            Ok(Some(Tile {
                first_read: r.first_read as usize,
                text: std::slice::from_raw_parts(r.text as *const u8, r.text_len as usize).to_vec(),
                text_off: std::slice::from_raw_parts(r.text_off, n + 1).to_vec(),
                status: std::slice::from_raw_parts(r.status, n).to_vec(),
                score: std::slice::from_raw_parts(r.score, n).to_vec(),
                device: r.device,
            }))
        }
    }
}

impl Drop for Stream {
    fn drop(&mut self) {
        unsafe { rg_stream_destroy(self.raw) }
    }
}
