/* Plain-C caller of the C ABI (include/recgraph_hip.h): no Python, no C++ types.  Reads a GFA and a FASTA
 * (rg_reads_from_fasta), aligns every read in mode argv[3] through the streaming engine (rg_stream_*: every visible device,
 * tiles in input order) and prints the GAF text — what the reference's `recgraph reads.fa graph.gfa -m <mode>` prints on
 * stdout.  argv[4] (optional): reads per tile.  Exit status: 0 ok, 3 no HIP device (RG_ERR_NO_DEVICE), 1 other.
 *     gcc -std=c11 -Iinclude tests/c/abi_smoke.c -Lrecgraph_amd -lrecgraph_hip -Wl,-rpath,recgraph_amd -o abi_smoke */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "recgraph_hip.h"

static char* slurp(const char* path, long* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    *len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* b = (char*)malloc((size_t)*len + 1);
    if (fread(b, 1, (size_t)*len, f) != (size_t)*len) { perror("fread"); exit(1); }
    b[*len] = 0;
    fclose(f);
    return b;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s graph.gfa reads.fa mode [tile]\n", argv[0]); return 1; }
    long glen, flen;
    char* gfa = slurp(argv[1], &glen);
    char* fa = slurp(argv[2], &flen);
    rg_reads* reads = NULL;
    if (rg_reads_from_fasta(fa, flen, &reads) != RG_OK) { fprintf(stderr, "fasta: %s\n", rg_last_error()); return 1; }
    rg_graph* g = NULL;
    if (rg_graph_from_gfa(gfa, glen, &g) != RG_OK) { fprintf(stderr, "graph: %s\n", rg_last_error()); return 1; }
    rg_params p;
    rg_params_default(&p, atoi(argv[3]));
    rg_stream_opts o;
    rg_stream_opts_default(&o);
    o.keep_records = 1;
    if (argc > 4) o.tile_reads = atoi(argv[4]);
    rg_stream* s = NULL;
    int rc = rg_stream_create(g, &p, NULL, 0, &o, &s);
    if (rc != RG_OK) {
        fprintf(stderr, "align: %d %s\n", rc, rg_last_error());
        rg_graph_destroy(g);
        return rc == RG_ERR_NO_DEVICE ? 3 : 1;
    }
    if (rg_stream_push(s, rg_reads_bases(reads), rg_reads_offsets(reads), rg_reads_count(reads), rg_reads_names(reads)) != RG_OK ||
        rg_stream_finish(s) != RG_OK) {
        fprintf(stderr, "push: %s\n", rg_last_error());
        return 1;
    }
    int tiles = 0;
    for (;;) {
        rg_stream_result r;
        rc = rg_stream_next(s, &r);
        if (rc == RG_STREAM_END) break;
        if (rc != RG_OK) { fprintf(stderr, "align: %d %s\n", rc, rg_last_error()); return 1; }
        fwrite(r.text, 1, (size_t)r.text_len, stdout);
        if (tiles++ == 0) {
            /* structured record of the first read, through the tile's results-only handle */
            rg_gaf_fields f;
            if (rg_result_fields(r.records, 0, &f, NULL, 0, NULL, 0) != RG_OK) return 1;
            fprintf(stderr, "read 0: has_record %d, query_length %llu, path ids %lld, comments %lld bytes\n", f.has_record,
                    (unsigned long long)f.query_length, (long long)f.n_path_ids, (long long)f.comments_len);
        }
    }
    fprintf(stderr, "%d tile(s), %d handle(s)\n", tiles, rg_stream_handles(s));
    rg_stream_destroy(s);
    rg_reads_destroy(reads);
    rg_graph_destroy(g);
    free(gfa);
    free(fa);
    return 0;
}
