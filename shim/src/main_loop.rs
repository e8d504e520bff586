//! The read loops of RecGraph's `src/main.rs` (:56-105 mode 0, :107-172 mode 1, :174-213 mode 2, :215-253 mode 3,
//! :255-262 mode 4, :263-270 mode 5, :289-313 modes 8 / 9) over the MI355X library: FASTA in, GAF out, every read of the
//! file through ONE `hip::Stream` (all visible GPUs).  A maintainer calls `align_all` from `main()` in place of the
//! `match align_mode { .. }` block; `-s true` (the reverse-complement retry of modes 0-3, main.rs:82-106, 132-165,
//! 188-212, 229-253) runs inside the library (`rg_stream_opts.amb_strand`).  The FASTA file is fed block by block and the
//! stream is bounded (`max_queued_tiles`, `max_undelivered_bytes`): like the reference's loop, memory does not grow with
//! the read set.
//! UNCOMPILED: no Rust toolchain exists in the image this was written in.
use std::io::Write;

use crate::args_parser;
use crate::hip;
use crate::utils;

/// `-m` of the command line -> `rg_params.mode` (args_parser.rs:31-38).
fn mode_of(align_mode: i32) -> i32 {
    match align_mode {
        0 => hip::RG_MODE_GLOBAL_POA,
        1 => hip::RG_MODE_LOCAL_POA,
        2 => hip::RG_MODE_GAP_POA,
        3 => hip::RG_MODE_GAP_LOCAL_POA,
        4 => hip::RG_MODE_PATHWISE,
        5 => hip::RG_MODE_PATHWISE_SEMI,
        8 => hip::RG_MODE_RECOMBINATION,
        9 => hip::RG_MODE_RECOMBINATION_SEMI,
        _ => panic!("Alignment mode must be in [0..9]"), // main.rs:315-317 (6 and 7 stay on the CPU path)
    }
}

/// Parameters of the run from the same getters `main.rs` uses (args_parser.rs:148-202).
fn params_of(align_mode: i32, score_matrix: &std::collections::HashMap<(char, char), i32>) -> hip::rg_params {
    let mut p = hip::default_params(mode_of(align_mode));
    let alphabet = ['A', 'C', 'G', 'T', 'N', '-'];
    for (a, ca) in alphabet.iter().enumerate() {
        for (b, cb) in alphabet.iter().enumerate() {
            p.scores[a * 6 + b] = *score_matrix.get(&(*ca, *cb)).unwrap_or(&hip::RG_SCORE_MISSING);
        }
    }
    let (b, f) = args_parser::get_b_f();
    p.band_b = b;
    p.band_f = f;
    let (o, e) = args_parser::get_gap_open_gap_ext();
    p.gap_open = o;
    p.gap_ext = e;
    let (base_rec_cost, multi_rec_cost) = args_parser::get_base_multi_recombination_cost();
    p.base_rec_cost = base_rec_cost;
    p.multi_rec_cost = multi_rec_cost;
    p.rec_band_width = args_parser::get_recombination_band_width();
    p
}

/// Aligns every read of `sequence_path` against `graph_path` in mode `align_mode` and writes what `main.rs` writes:
/// the warning lines on stdout, every record through `utils::write_gaf` (stdout or the `-o` file, with the number
/// `main.rs` passes: `i + 1` in modes 0-3, `i` in modes 4, 5, 8, 9).
pub fn align_all(align_mode: i32, sequence_path: &str, graph_path: &str, score_matrix: &std::collections::HashMap<(char, char), i32>) {
    use std::io::Read;
    let gfa = std::fs::read_to_string(graph_path).unwrap();
    let graph = hip::Graph::from_gfa_text(&gfa).unwrap_or_else(|e| panic!("{}", e));
    let params = params_of(align_mode, score_matrix);
    let mut opts = hip::Stream::default_opts();
    opts.amb_strand = args_parser::get_amb_strand_mode() as i32; // ignored by modes 4+ like main.rs:254-313
    // bounded like the reference's loop (one read in, one record out): pushes wait while 4 tiles are queued, the workers
    // wait while 64 MB of finished text has not been taken; a feeder thread reads the file block by block (a push may
    // wait, so feeding and draining must not be the same thread)
    opts.max_queued_tiles = 4;
    opts.max_undelivered_bytes = 64 << 20;
    let stream = hip::Stream::new(&graph, &params, None, Some(opts)).unwrap_or_else(|e| panic!("{}", e));
    // The reference parses the whole file before it aligns anything and panics on a malformed one (sequences.rs:41-43): the
    // same check runs over the file (counts only) before the first byte of output.
    hip::fasta_check_file(sequence_path).unwrap_or_else(|e| panic!("{}", e));
    let stdout = std::io::stdout();
    // The feeder NEVER panics and ALWAYS closes the stream: a panic inside a scoped thread is only seen at the join, while
    // the main thread would sit in `next()` forever; and a main thread that unwinds while the feeder waits inside a bounded
    // push would wait for that join forever.  So: the feeder hands its error over, and every error path of the main
    // thread aborts the stream (which wakes a blocked push) before it leaves the scope.
    let result: Result<(), String> = std::thread::scope(|sc| {
        let feeder = sc.spawn(|| -> Result<(), String> {
            let fed = (|| -> Result<(), String> {
                let mut file = std::fs::File::open(sequence_path).map_err(|e| e.to_string())?;
                let mut block = vec![0u8; 4 << 20];
                loop {
                    // sequences::get_sequences (sequences.rs:5-45) runs inside the library
                    let got = file.read(&mut block).map_err(|e| e.to_string())?;
                    stream.feed_fasta(&block[..got], got == 0)?;
                    if got == 0 {
                        return Ok(());
                    }
                }
            })();
            if fed.is_err() {
                stream.abort(); // the consumer must not wait for tiles that will never come
            }
            let _ = stream.finish();
            fed
        });
        let drained = (|| -> Result<(), String> {
            while let Some(tile) = stream.next()? {
                for i in 0..tile.status.len() {
                    if tile.status[i] & (hip::RG_READ_WOULD_PANIC | hip::RG_READ_BAD_BASE) != 0 {
                        return Err(format!("read {}: the CPU path panics on this input", tile.first_read + i));
                    }
                    let text = &tile.text[tile.text_off[i] as usize..tile.text_off[i + 1] as usize];
                    // the last line is the GAF record; lines before it are the `println!` warnings of the exec functions
                    let body = &text[..text.len() - 1];
                    let cut = body.iter().rposition(|c| *c == b'\n').map(|p| p + 1).unwrap_or(0);
                    stdout.lock().write_all(&text[..cut]).map_err(|e| e.to_string())?;
                    let record = String::from_utf8_lossy(&body[cut..]).into_owned();
                    let n = tile.first_read + i;
                    utils::write_gaf(&record, if align_mode <= 3 { n + 1 } else { n });
                }
            }
            Ok(())
        })();
        if drained.is_err() {
            stream.abort(); // wakes a feeder blocked in a bounded push: the scope can join it
        }
        let fed = feeder.join().unwrap_or_else(|_| Err("feeder thread panicked".to_string()));
        // the feeder's error is the cause when both failed ("stream aborted" is only its echo on the other side)
        fed.and(drained)
    });
    if let Err(e) = result {
        panic!("{}", e);
    }
}
