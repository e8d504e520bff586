//! Drop-in replacement of RecGraph's `src/api.rs`: the same public functions and signatures (`api.rs:11,43,76,102,131,153`)
//! computing on the MI355X through `crate::hip` (C ABI `include/recgraph_hip.h`).
//! UNCOMPILED: no Rust toolchain exists in the image this was written in.
use std::collections::HashMap;

use crate::gaf_output::GAFStruct;
use crate::hip;
use crate::score_matrix;
use handlegraph::{
    handle::{Direction, Handle},
    handlegraph::HandleGraph,
    hashgraph::HashGraph,
};

const ALPHABET: [char; 6] = ['A', 'C', 'G', 'T', 'N', '-'];

/// `HashGraph` -> flat arrays with the rules of `graph::create_graph_struct` (graph.rs:31-123): handles sorted by id,
/// bases concatenated between '$' and 'F'; the first row of a segment lists the last rows of its left neighbours in
/// `handle_edges_iter(.., Left)` order (row 0 when it has none); row F lists the last row of every segment without a
/// right neighbour — in ascending row order here (the reference iterates a HashMap, graph.rs:112-123).
struct Flat {
    lnz: Vec<u8>,
    pred_off: Vec<i64>,
    pred_rows: Vec<i64>,
    node_id: Vec<u64>,
}

fn flatten(graph: &HashGraph) -> Flat {
    let mut handles: Vec<Handle> = graph.handles_iter().collect();
    handles.sort();
    let mut lnz: Vec<u8> = vec![b'$'];
    let mut node_id: Vec<u64> = vec![0];
    let mut last_row: HashMap<u64, i64> = HashMap::new();
    let mut first_row: Vec<i64> = Vec::with_capacity(handles.len());
    for h in &handles {
        first_row.push(lnz.len() as i64);
        for c in graph.sequence(*h) {
            lnz.push(c);
            node_id.push(u64::from(h.id()));
        }
        last_row.insert(u64::from(h.id()), lnz.len() as i64 - 1);
    }
    lnz.push(b'F');
    node_id.push(0);
    let rows = lnz.len();
    let mut preds: Vec<Vec<i64>> = vec![Vec::new(); rows];
    let mut has_right: HashMap<u64, bool> = HashMap::new();
    for (k, h) in handles.iter().enumerate() {
        let start = first_row[k] as usize;
        let mut any = false;
        for p in graph.handle_edges_iter(*h, Direction::Left) {
            any = true;
            preds[start].push(last_row[&u64::from(p.id())]);
            has_right.insert(u64::from(p.id()), true);
        }
        if !any {
            preds[start].push(0);
        }
    }
    for h in &handles {
        if !has_right.contains_key(&u64::from(h.id())) {
            preds[rows - 1].push(last_row[&u64::from(h.id())]);
        }
    }
    preds[rows - 1].sort();
    let mut pred_off: Vec<i64> = Vec::with_capacity(rows + 1);
    let mut pred_rows: Vec<i64> = Vec::new();
    pred_off.push(0);
    for p in &preds {
        pred_rows.extend_from_slice(p);
        pred_off.push(pred_rows.len() as i64);
    }
    Flat { lnz, pred_off, pred_rows, node_id }
}

fn table_i32(m: &HashMap<(char, char), i32>) -> [i32; 36] {
    let mut t = [hip::RG_SCORE_MISSING; 36];
    for (a, ca) in ALPHABET.iter().enumerate() {
        for (b, cb) in ALPHABET.iter().enumerate() {
            if let Some(v) = m.get(&(*ca, *cb)) {
                t[a * 6 + b] = *v;
            }
        }
    }
    t
}

fn table_f32(m: &HashMap<(char, char), f32>) -> [i32; 36] {
    // exec_simd computes in f32 on integer-valued scores (exact below 2^24): the library takes them as integers
    let mut t = [hip::RG_SCORE_MISSING; 36];
    for (a, ca) in ALPHABET.iter().enumerate() {
        for (b, cb) in ALPHABET.iter().enumerate() {
            if let Some(v) = m.get(&(*ca, *cb)) {
                assert!(v.fract() == 0.0, "non-integer scores are outside the exact range of the f32 path");
                t[a * 6 + b] = *v as i32;
            }
        }
    }
    t
}

fn run(read: &String, graph: &HashGraph, sequence_name: Option<(&str, usize)>, params: hip::rg_params) -> GAFStruct {
    let (name, index) = sequence_name.unwrap_or(("no_name", 1));
    // seq_name.1 == 0 asks the exec functions for the score only; the reference then unwraps a None
    assert!(index != 0, "called `Option::unwrap()` on a `None` value");
    let flat = flatten(graph);
    let g = hip::Graph::from_lnz(&flat.lnz, &flat.pred_off, &flat.pred_rows, &flat.node_id).expect("rg_graph_create_lnz");
    let b = hip::Batch::align(&g, &params, &[read.as_str()]).expect("rg_align_batch");
    let rec = b.record(0).expect("rg_result_fields");
    if rec.status & (hip::RG_READ_WOULD_PANIC | hip::RG_READ_BAD_BASE) != 0 || rec.fields.has_record == 0 {
        panic!("the reference panics on this input (status bits {})", rec.status);
    }
    // the lines the exec functions println! before returning
    if rec.fields.warning & hip::RG_READ_BAND_WARNING != 0 {
        println!("Band length probably too short, maybe try with larger b and f");
    }
    if rec.fields.warning & hip::RG_READ_BAND_NOT_ENOUGH != 0 {
        println!("band not enough for correct output");
    }
    if rec.fields.empty != 0 {
        return GAFStruct::new();
    }
    GAFStruct::build_gaf_struct(
        String::from(name),
        rec.fields.query_length as usize,
        rec.fields.query_start as usize,
        rec.fields.query_end as usize,
        rec.fields.strand as u8 as char,
        rec.path,
        rec.fields.path_length as usize,
        rec.fields.path_start as usize,
        rec.fields.path_end as usize,
        rec.fields.residue_matches_number as usize,
        String::from("*"),
        String::from("*"),
        rec.comments,
    )
}

fn bases_to_add(read: &String, bases_to_add: Option<f32>) -> i64 {
    (read.len() as f32 * bases_to_add.unwrap_or(0.1)) as usize as i64
}

/// Global alignment with adaptive band (the exec_simd semantics of the reference), score matrix can be set with
/// create_score_matrix_f32.  Only required parameters are a read as a &String and a graph as a &HandleGraph.
pub fn align_global_no_gap(
    read: &String,
    graph: &HashGraph,
    sequence_name: Option<(&str, usize)>,
    score_matrix: Option<HashMap<(char, char), f32>>,
    bases_to_add_frac: Option<f32>,
) -> GAFStruct {
    let m = score_matrix.unwrap_or(score_matrix::create_score_matrix_match_mis_f32(2f32, -4f32));
    let mut p = hip::default_params(hip::RG_MODE_GLOBAL_POA);
    p.scores = table_f32(&m);
    p.bta_override = bases_to_add(read, bases_to_add_frac);
    run(read, graph, sequence_name, p)
}

/// Global alignment with adaptive band and affine gaps, score matrix can be set with create_score_matrix_i32.
pub fn align_global_gap(
    read: &String,
    graph: &HashGraph,
    sequence_name: Option<(&str, usize)>,
    score_matrix: Option<HashMap<(char, char), i32>>,
    bases_to_add_frac: Option<f32>,
    o: Option<i32>,
    e: Option<i32>,
) -> GAFStruct {
    let m = score_matrix.unwrap_or(score_matrix::create_score_matrix_match_mis(2, -4));
    let mut p = hip::default_params(hip::RG_MODE_GAP_POA);
    p.scores = table_i32(&m);
    p.bta_override = bases_to_add(read, bases_to_add_frac);
    p.gap_open = o.unwrap_or(-10);
    p.gap_ext = e.unwrap_or(-6);
    run(read, graph, sequence_name, p)
}

/// Local alignment (the exec_simd semantics of the reference), score matrix can be set with create_score_matrix_f32.
pub fn align_local_no_gap(
    read: &String,
    graph: &HashGraph,
    sequence_name: Option<(&str, usize)>,
    score_matrix: Option<HashMap<(char, char), f32>>,
) -> GAFStruct {
    let m = score_matrix.unwrap_or(score_matrix::create_score_matrix_match_mis_f32(2f32, -4f32));
    let mut p = hip::default_params(hip::RG_MODE_LOCAL_POA);
    p.scores = table_f32(&m);
    run(read, graph, sequence_name, p)
}

/// Local gap alignment, score matrix can be set with create_score_matrix_i32.
pub fn align_local_gap(
    read: &String,
    graph: &HashGraph,
    sequence_name: Option<(&str, usize)>,
    score_matrix: Option<HashMap<(char, char), i32>>,
    o: Option<i32>,
    e: Option<i32>,
) -> GAFStruct {
    let m = score_matrix.unwrap_or(score_matrix::create_score_matrix_match_mis(2, -4));
    let mut p = hip::default_params(hip::RG_MODE_GAP_LOCAL_POA);
    p.scores = table_i32(&m);
    p.gap_open = o.unwrap_or(-10);
    p.gap_ext = e.unwrap_or(-6);
    run(read, graph, sequence_name, p)
}

/// Returns a score matrix for gap alignments, can be set with match/mismatch score or by parsing a .mtx file
/// (host-side table building stays with the crate's own score_matrix module).
pub fn create_score_matrix_i32(
    match_score: Option<i32>,
    mismatch_score: Option<i32>,
    matrix_file_path: Option<&str>,
) -> HashMap<(char, char), i32> {
    match matrix_file_path {
        Some(path) => score_matrix::create_score_matrix_from_matrix_file(path),
        _ => score_matrix::create_score_matrix_match_mis(match_score.unwrap(), mismatch_score.unwrap()),
    }
}

/// Returns a score matrix for the no-gap alignments (f32 values of the i32 matrix above).
pub fn create_score_matrix_f32(
    match_score: Option<i32>,
    mismatch_score: Option<i32>,
    matrix_type: Option<&str>,
) -> HashMap<(char, char), f32> {
    create_score_matrix_i32(match_score, mismatch_score, matrix_type)
        .iter()
        .map(|(k, v)| (*k, *v as f32))
        .collect()
}
