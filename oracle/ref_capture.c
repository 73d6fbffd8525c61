/*
 * ref_capture — TEST INFRASTRUCTURE ONLY (oracle side, never linked into the product).
 *
 * Runs the *unmodified* reference mTR (compiled from /root/reference into
 * oracle/_ref/libmtr_ref.so by oracle/Makefile) and records what it computes at the
 * capture points of SURVEY.md §8(c), without touching a single reference source line:
 * the reference is built -fPIC, so every global function call goes through the PLT and
 * the definitions in this executable interpose them; each hook forwards to the real
 * function found with dlsym(RTLD_NEXT).
 *
 *   G1  after fill_directional_index_with_end   (fill_directional_index.c:549-602)
 *   G2  per search_De_Bruijn_graph call         (consensus.c:507-582)
 *   G3  per wrap_around_DP_sub call             (wrap_around_DP.c:222-354)
 *   G3p per polish_repeat call                  (consensus.c:610-704)
 *   G3r per revise_representative_unit_sub call (consensus.c:851-1046)
 *   G4  per insert_an_alignment_into_set call   (chaining.cpp:203-241), insertion order
 *   G5  = the reference's own stdout (left untouched on fd 1)
 *
 * usage: ref_capture [-a] [-p] [-m ratio] [-l level] <fasta> <capture.jsonl>
 *   level 0: G4 only; 1: + G1 G3 G3p G3r; 2: + G2 (calls that pass maxFreq>5, i.e. that
 *   return a unit or ran a search); 3: + every G2 call.
 * Isolated semantics = run this once per read; tests/golden/make_golden.py
 * does the splitting.
 *
 * Built only where /root/reference exists (needs its mTR.h for the record layout).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include "mTR.h" /* from -I/root/reference: struct repeat_in_read + the globals */

static FILE *cap = NULL;
static int level = 1;

#define REAL(name) static __typeof__(&name) real = NULL; if (!real) { real = (__typeof__(&name))dlsym(RTLD_NEXT, #name); if (!real) { fprintf(stderr, "ref_capture: no real %s\n", #name); exit(2);} }

static void put_rr(const repeat_in_read *rr)
{
    fprintf(cap, "{\"rep_start\":%d,\"rep_end\":%d,\"repeat_len\":%d,\"period\":%d,\"copies\":%d,"
                 "\"mat\":%d,\"mis\":%d,\"ins\":%d,\"del\":%d,\"k\":%d,\"G\":%d,\"MM\":%d,\"D\":%d,\"unit\":\"",
            rr->rep_start, rr->rep_end, rr->repeat_len, rr->rep_period, rr->Num_freq_unit,
            rr->Num_matches, rr->Num_mismatches, rr->Num_insertions, rr->Num_deletions, rr->Kmer,
            rr->match_gain, rr->mismatch_penalty, rr->indel_penalty);
    if (rr->rep_period > 0 && rr->rep_period < MAX_PERIOD) fwrite(rr->string, 1, strnlen(rr->string, MAX_PERIOD), cap);
    fprintf(cap, "\",\"score\":[");
    if (rr->rep_period > 0 && rr->rep_period < MAX_PERIOD)
        for (int i = 0; i < rr->rep_period; i++) fprintf(cap, "%s%d", i ? "," : "", rr->string_score[i]);
    fprintf(cap, "]}");
}

void fill_directional_index_with_end(int DI_array_length, int inputLen, int random_string_length)
{
    REAL(fill_directional_index_with_end);
    real(DI_array_length, inputLen, random_string_length);
    if (level < 1) return;
    fprintf(cap, "{\"t\":\"G1\",\"L\":%d,\"r\":%d,\"ranges\":[", inputLen, random_string_length);
    int first = 1;
    for (int i = 0; i < inputLen; i++) {
        if (directional_index[i] != -1 || directional_index_end[i] > -1) {
            uint64_t bits; memcpy(&bits, &directional_index[i], 8);
            fprintf(cap, "%s[%d,%d,%d,\"%016llx\"]", first ? "" : ",", i, directional_index_end[i],
                    directional_index_w[i], (unsigned long long)bits);
            first = 0;
        }
    }
    fprintf(cap, "]}\n");
}

int search_De_Bruijn_graph(int query_start, int query_end, repeat_in_read *rr)
{
    REAL(search_De_Bruijn_graph);
    int k = rr->Kmer;
    int found = real(query_start, query_end, rr);
    if (level >= 3 || (level >= 2 && (found || rr->rep_period != -1))) {
        fprintf(cap, "{\"t\":\"G2\",\"qs\":%d,\"qe\":%d,\"k\":%d,\"found\":%d,\"rr\":", query_start, query_end, k, found);
        put_rr(rr);
        fprintf(cap, "}\n");
    }
    return found;
}

void wrap_around_DP_sub(int query_start, int query_end, repeat_in_read *rr, int G, int MM, int D)
{
    REAL(wrap_around_DP_sub);
    char unit[MAX_PERIOD + 1];
    int U = rr->rep_period;
    if (U < 0) U = 0; if (U > MAX_PERIOD) U = MAX_PERIOD;
    memcpy(unit, rr->string, U); unit[U] = 0;
    real(query_start, query_end, rr, G, MM, D);
    if (level < 1) return;
    fprintf(cap, "{\"t\":\"G3\",\"qs\":%d,\"qe\":%d,\"unit\":\"%s\",\"G\":%d,\"MM\":%d,\"D\":%d,\"out\":[%d,%d,%d,%d,%d,%d,%d,%d]}\n",
            query_start, query_end, unit, G, MM, D, rr->rep_start, rr->rep_end, rr->repeat_len,
            rr->Num_freq_unit, rr->Num_matches, rr->Num_mismatches, rr->Num_insertions, rr->Num_deletions);
}

void polish_repeat(repeat_in_read *rr)
{
    REAL(polish_repeat);
    char unit[MAX_PERIOD + 1];
    int U = rr->rep_period; if (U < 0) U = 0; if (U > MAX_PERIOD) U = MAX_PERIOD;
    memcpy(unit, rr->string, U); unit[U] = 0;
    int rs = rr->rep_start, re = rr->rep_end, k = rr->Kmer;
    real(rr);
    if (level < 1) return;
    char out[MAX_PERIOD + 1];
    int V = rr->rep_period; if (V < 0) V = 0; if (V > MAX_PERIOD) V = MAX_PERIOD;
    memcpy(out, rr->string, V); out[V] = 0;
    fprintf(cap, "{\"t\":\"G3p\",\"rep_start\":%d,\"rep_end\":%d,\"k\":%d,\"in\":\"%s\",\"in_score\":[", rs, re, k, unit);
    for (int i = 0; i < U; i++) fprintf(cap, "%s%d", i ? "," : "", rr->string_score[i]);
    fprintf(cap, "],\"out\":\"%s\"}\n", out);
}

void revise_representative_unit_sub(repeat_in_read *rr, int G, int MM, int D)
{
    REAL(revise_representative_unit_sub);
    char unit[MAX_PERIOD + 1];
    int U = rr->rep_period; if (U < 0) U = 0; if (U > MAX_PERIOD) U = MAX_PERIOD;
    memcpy(unit, rr->string, U); unit[U] = 0;
    int rs = rr->rep_start, re = rr->rep_end, rl = rr->repeat_len;
    int mis = rr->Num_mismatches, ins = rr->Num_insertions, del = rr->Num_deletions;
    real(rr, G, MM, D);
    if (level < 1) return;
    fprintf(cap, "{\"t\":\"G3r\",\"rep_start\":%d,\"rep_end\":%d,\"repeat_len\":%d,\"mis\":%d,\"ins\":%d,\"del\":%d,"
                 "\"G\":%d,\"MM\":%d,\"D\":%d,\"in\":\"%s\",\"out_period\":%d,\"out\":\"",
            rs, re, rl, mis, ins, del, G, MM, D, unit, rr->rep_period);
    /* the revised unit may be up to 2x MAX_PERIOD long in principle; string[] holds it NUL-terminated */
    fputs(rr->string, cap);
    fprintf(cap, "\"}\n");
}

void insert_an_alignment_into_set(char *readID, int inputLen, int rep_start, int rep_end, int repeat_len,
                                  int rep_period, int Num_freq_unit, int Num_matches, int Num_mismatches,
                                  int Num_insertions, int Num_deletions, int Kmer, int match_gain,
                                  int mismatch_penalty, int indel_penalty, char *string, int *string_score)
{
    REAL(insert_an_alignment_into_set);
    fprintf(cap, "{\"t\":\"G4\",\"id\":\"%s\",\"L\":%d,\"rep_start\":%d,\"rep_end\":%d,\"repeat_len\":%d,\"period\":%d,"
                 "\"copies\":%d,\"mat\":%d,\"mis\":%d,\"ins\":%d,\"del\":%d,\"k\":%d,\"G\":%d,\"MM\":%d,\"D\":%d,\"unit\":\"%s\",\"score\":[",
            readID, inputLen, rep_start, rep_end, repeat_len, rep_period, Num_freq_unit, Num_matches,
            Num_mismatches, Num_insertions, Num_deletions, Kmer, match_gain, mismatch_penalty, indel_penalty, string);
    for (int i = 0; i < rep_period; i++) fprintf(cap, "%s%d", i ? "," : "", string_score[i]);
    fprintf(cap, "]}\n");
    real(readID, inputLen, rep_start, rep_end, repeat_len, rep_period, Num_freq_unit, Num_matches, Num_mismatches,
         Num_insertions, Num_deletions, Kmer, match_gain, mismatch_penalty, indel_penalty, string, string_score);
}

int main(int argc, char **argv)
{
    int print_alignment = 0;
    Manhattan_Distance = 1;
    min_match_ratio = MIN_MATCH_RATIO;
    int opt;
    while ((opt = getopt(argc, argv, "apm:l:")) != -1) {
        switch (opt) {
        case 'a': print_alignment = 1; break;
        case 'p': Manhattan_Distance = 0; break;
        case 'm': min_match_ratio = atof(optarg); break;
        case 'l': level = atoi(optarg); break;
        default: fprintf(stderr, "usage: ref_capture [-a] [-p] [-m r] [-l level] fasta capture.jsonl\n"); return 2;
        }
    }
    if (optind + 2 > argc) { fprintf(stderr, "usage: ref_capture [-a] [-p] [-m r] [-l level] fasta capture.jsonl\n"); return 2; }
    cap = fopen(argv[optind + 1], "w");
    if (!cap) { perror("capture file"); return 2; }
    time_all = time_memory = time_range = time_period = 0;
    time_initialize_input_string = time_count_table = time_wrap_around_DP = time_chaining = 0;
    query_counter = 0;
    handle_one_file(argv[optind], print_alignment);
    fclose(cap);
    return 0;
}
