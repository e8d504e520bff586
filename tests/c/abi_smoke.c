/* Plain-C caller of the C ABI (include/recgraph_hip.h): no Python, no C++ types.  Reads a GFA and a FASTA, aligns every
 * read in mode argv[3] through rg_align_batch_multi on all visible devices and prints the GAF text — what the reference's
 * `recgraph reads.fa graph.gfa -m <mode>` prints on stdout.  Exit status: 0 ok, 3 no HIP device (RG_ERR_NO_DEVICE), 1 other.
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
    if (argc < 4) { fprintf(stderr, "usage: %s graph.gfa reads.fa mode\n", argv[0]); return 1; }
    long glen, flen;
    char* gfa = slurp(argv[1], &glen);
    char* fa = slurp(argv[2], &flen);
    /* FASTA: '>' header lines name the reads, the other lines are bases (sequences.rs:5-45) */
    char* reads = (char*)malloc((size_t)flen + 1);
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * ((size_t)flen / 2 + 2));
    const char** names = (const char**)malloc(sizeof(char*) * ((size_t)flen / 2 + 2));
    int64_t n = 0, pos = 0;
    for (char* line = strtok(fa, "\n"); line; line = strtok(NULL, "\n")) {
        size_t l = strlen(line);
        if (l && line[l - 1] == '\r') line[--l] = 0;
        if (line[0] == '>') { off[n] = pos; names[n++] = line + 1; }
        else { memcpy(reads + pos, line, l); pos += (int64_t)l; }
    }
    off[n] = pos;
    rg_graph* g = NULL;
    if (rg_graph_from_gfa(gfa, glen, &g) != RG_OK) { fprintf(stderr, "graph: %s\n", rg_last_error()); return 1; }
    rg_params p;
    rg_params_default(&p, atoi(argv[3]));
    rg_multi* m = NULL;
    int rc = rg_align_batch_multi(g, &p, reads, off, n, NULL, 0, &m);
    if (rc != RG_OK) {
        fprintf(stderr, "align: %d %s\n", rc, rg_last_error());
        rg_graph_destroy(g);
        return rc == RG_ERR_NO_DEVICE ? 3 : 1;
    }
    int64_t need = rg_multi_format_all(m, names, 1, NULL, 0, 4);
    char* text = (char*)malloc((size_t)need + 1);
    rg_multi_format_all(m, names, 1, text, need + 1, 4);
    fwrite(text, 1, (size_t)need, stdout);
    /* structured record of the first read, through the shard that holds it */
    rg_gaf_fields f;
    if (rg_result_fields(rg_multi_batch(m, 0), 0, &f, NULL, 0, NULL, 0) != RG_OK) return 1;
    fprintf(stderr, "read 0: has_record %d, query_length %llu, path ids %lld, comments %lld bytes, %d shard(s)\n", f.has_record,
            (unsigned long long)f.query_length, (long long)f.n_path_ids, (long long)f.comments_len, rg_multi_shards(m));
    rg_multi_destroy(m);
    rg_graph_destroy(g);
    return 0;
}
