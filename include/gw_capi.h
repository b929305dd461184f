/*
 * gw_capi.h -- object-level C-ABI over the host C++ classes (libgenomeworks_amd.so).
 *
 * These are the entry points a foreign-function binding of the reference's public API binds:
 * the reference exposes cudapoa::Batch / cudaaligner::Aligner to Python through Cython
 * (pygenomeworks/genomeworks/cudapoa/cudapoa.pxd:36-110, cudaaligner/cudaaligner.pxd:36-100);
 * every function here flattens one method of those classes (same argument meaning, same status codes).
 * Plain pointers and sizes only; strings are returned through caller-provided buffers or
 * library-owned buffers that stay valid until the next call on the same handle.
 */
#ifndef GW_CAPI_H
#define GW_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gw_poa_batch gw_poa_batch;
typedef struct gw_aligner gw_aligner;

/* last error (exception text) of the calling thread */
const char* gw_last_error(void);

/* ---- device / runtime helpers (pygenomeworks/genomeworks/cuda/cuda.pyx) ---- */
int gw_device_count(int* count);
int gw_set_device(int device);
int gw_get_device(int* device);
int gw_mem_info(size_t* free_bytes, size_t* total_bytes);
int gw_stream_create(void** stream);
int gw_stream_sync(void* stream);
int gw_stream_destroy(void* stream);

/* ---- synthetic inputs (genomeutils.hpp:32-127 semantics; std::minstd_rand(seed)) ---- */
/* Window = backbone of `backbone_len` random ACGT + (n_reads-1) x generate_random_sequence(backbone, rng,
   max_mut, max_ins, max_del). Reads are written back to back (no padding) into `out` (capacity out_cap),
   lengths into lens[n_reads]. Returns total bytes written, or -1 if out_cap is too small. */
int64_t gw_generate_window(uint32_t seed, int32_t backbone_len, int32_t n_reads, int32_t max_mut, int32_t max_ins,
                           int32_t max_del, char* out, int64_t out_cap, int32_t* lens);
/* Aligner pair: query = random genome of `len`; target = generate_random_sequence(query, rng, mut, ins, del);
   one rng stream (seeded once per call with `seed`) across `n_pairs` pairs, as cudaaligner/benchmarks/main.cpp:116-127. */
int64_t gw_generate_pairs(uint32_t seed, int32_t n_pairs, int32_t len, int32_t max_mut, int32_t max_ins,
                          int32_t max_del, char* out, int64_t out_cap, int32_t* qlens, int32_t* tlens);

/* The random test pairs of the reference's aligner tests (cudaaligner/tests/cudaaligner_test_cases.cpp:29-41): per pair
   a target of uniform random length in [0, max_len], then query = generate_random_sequence(target, rng, len, len, len);
   one std::minstd_rand(seed) stream. Written as t0 q0 t1 q1 ...; returns total bytes, -1 if out_cap is too small. */
int64_t gw_generate_random_length_pairs(uint32_t seed, int32_t n_pairs, int32_t max_len, char* out, int64_t out_cap,
                                        int32_t* tlens, int32_t* qlens);

/* ---- cudapoa::BatchConfig (batch.hpp:60-86, batch.cu:34-104) ---- */
typedef struct gw_poa_batch_config
{
    int32_t max_sequence_size;
    int32_t max_consensus_size;
    int32_t max_nodes_per_graph;
    int32_t matrix_sequence_dimension;
    int32_t alignment_band_width;
    int32_t max_sequences_per_poa;
    int32_t band_mode;
    int32_t max_banded_pred_distance;
} gw_poa_batch_config;

/* BatchConfig(max_seq_sz, max_seq_per_poa, band_width, banding, adaptive_storage_factor, graph_length_factor, max_pred_dist) */
int gw_poa_batch_config_default(gw_poa_batch_config* out, int32_t max_seq_sz, int32_t max_seq_per_poa,
                                int32_t band_width, int32_t band_mode, float adaptive_storage_factor,
                                float graph_length_factor, int32_t max_pred_dist);
/* BatchConfig(max_seq_sz, max_consensus_sz, max_nodes_per_poa, band_width, max_seq_per_poa, matrix_seq_dim, banding, max_pred_distance) */
int gw_poa_batch_config_full(gw_poa_batch_config* out, int32_t max_seq_sz, int32_t max_consensus_sz,
                             int32_t max_nodes_per_poa, int32_t band_width, int32_t max_seq_per_poa,
                             int32_t matrix_seq_dim, int32_t band_mode, int32_t max_pred_distance);

/* create_batch(device_id, stream, max_mem, output_mask, batch_size, gap, mismatch, match)  batch.hpp:194-204.
   Returns NULL on exception (see gw_last_error). */
gw_poa_batch* gw_poa_create_batch(int32_t device_id, void* stream, int64_t max_mem, int8_t output_mask,
                                  const gw_poa_batch_config* cfg, int16_t gap_score, int16_t mismatch_score,
                                  int16_t match_score);
void gw_poa_destroy_batch(gw_poa_batch* b);

/* Batch::add_poa_group: seqs[n] pointers, weights[n] pointers (entries or the array itself may be NULL),
   lengths[n]; per_seq_status[n] receives the per-entry StatusType. Returns the group StatusType, -1 on exception. */
int gw_poa_add_poa_group(gw_poa_batch* b, int32_t n, const char* const* seqs, const int8_t* const* weights,
                         const int32_t* lengths, int32_t* per_seq_status);
int32_t gw_poa_get_total_poas(gw_poa_batch* b);
int gw_poa_generate_poa(gw_poa_batch* b);
int32_t gw_poa_batch_id(gw_poa_batch* b);
int gw_poa_reset(gw_poa_batch* b);
int32_t gw_poa_max_poas(gw_poa_batch* b);

/* Batch::get_consensus (cudapoa/include/.../batch.hpp:83-86) the way the reference's benchmark calls it
   (cudapoa/benchmarks/single_batch.hpp:86-93): three fresh vectors through the public virtual call; the handle then owns
   them (the previous call's are destroyed). Returns the StatusType of the call (output_type_unavailable = 9), -1 on
   exception; on any failure the handle holds no results. */
int gw_poa_get_consensus(gw_poa_batch* b, int32_t* n_out);
/* EXTENSION, no counterpart in cudapoa::Batch: fetch into the storage of the handle's previous results (no heap traffic in
   a steady-state loop). Same return convention. */
int gw_poa_get_consensus_in_place(gw_poa_batch* b, int32_t* n_out);
const char* gw_poa_consensus_str(gw_poa_batch* b, int32_t poa, int32_t* length);
const uint16_t* gw_poa_consensus_coverage(gw_poa_batch* b, int32_t poa, int32_t* length);
int32_t gw_poa_output_status(gw_poa_batch* b, int32_t poa);

/* Batch::get_msa */
int gw_poa_get_msa(gw_poa_batch* b, int32_t* n_out);
int32_t gw_poa_msa_rows(gw_poa_batch* b, int32_t poa);
const char* gw_poa_msa_row(gw_poa_batch* b, int32_t poa, int32_t row, int32_t* length);

/* Batch::get_graphs: per window node labels + (src, sink, weight) triples */
int gw_poa_get_graphs(gw_poa_batch* b, int32_t* n_out);
int32_t gw_poa_graph_num_nodes(gw_poa_batch* b, int32_t poa);
int32_t gw_poa_graph_num_edges(gw_poa_batch* b, int32_t poa);
int gw_poa_graph_copy(gw_poa_batch* b, int32_t poa, char* node_labels, int32_t* edge_src, int32_t* edge_dst,
                      int32_t* edge_weight);

/* timing / accounting helpers for bench.py: device cell counters of the last generate_poa() */
int gw_poa_total_cells(gw_poa_batch* b, uint64_t* cells);
/* device pointers + args of the last generate_poa (so a benchmark can re-launch with inputs resident in HBM) */
int gw_poa_relaunch(gw_poa_batch* b);
/* same, timed with HIP events on the batch's stream (ms): graph-build kernel, then consensus/MSA kernel */
int gw_poa_relaunch_timed(gw_poa_batch* b, float* graph_build_ms, float* output_ms);
/* profiling aid: mean s_memtime ticks per window of {row table, NW forward, sink+traceback, graph merge, topsort, other} */
int gw_poa_profile_phases(gw_poa_batch* b, double* out6);
/* the same six counters for every window of the batch (window-major); returns the number of windows written, <0 on error */
int gw_poa_profile_phases_per_window(gw_poa_batch* b, uint64_t* out, int32_t capacity_windows);

/* ---- cudapoa/multi_device.hpp: windows over several devices / several batches per device (one host thread, stream
   and Batch per worker; windows pulled from a shared cursor; results by global window index; no collective) ---- */
typedef struct gw_poa_multi gw_poa_multi;
/* seqs / lengths: the reads of all windows back to back (reads_per_window[w] of them per window). devices[n_devices]:
   device id per worker group (an id may repeat: logical shards of one device). memory_per_device: bytes per entry of
   devices, -1 = 0.9 x free / entries naming the device. Returns NULL on exception (gw_last_error). */
gw_poa_multi* gw_poa_multi_device_run(int32_t n_windows, const int32_t* reads_per_window, const char* const* seqs,
                                      const int32_t* lengths, const gw_poa_batch_config* cfg, const int32_t* devices,
                                      int32_t n_devices, int32_t batches_per_device, int64_t memory_per_device, int8_t output_mask,
                                      int16_t gap_score, int16_t mismatch_score, int16_t match_score);
void gw_poa_multi_destroy(gw_poa_multi* h);
int32_t gw_poa_multi_launches(gw_poa_multi* h);
double gw_poa_multi_seconds(gw_poa_multi* h); /* workers' wall time: batch creation, filling, kernels, result unpacking */
double gw_poa_multi_seconds_after_creation(gw_poa_multi* h); /* size-class runs: filling + generate_poa() + get_*() (multi_batch.hpp:72-177) */
int32_t gw_poa_multi_status(gw_poa_multi* h, int32_t window);
int32_t gw_poa_multi_worker(gw_poa_multi* h, int32_t window);
const char* gw_poa_multi_consensus(gw_poa_multi* h, int32_t window, int32_t* length);
const uint16_t* gw_poa_multi_coverage(gw_poa_multi* h, int32_t window, int32_t* length);
int32_t gw_poa_multi_msa_rows(gw_poa_multi* h, int32_t window);
const char* gw_poa_multi_msa_row(gw_poa_multi* h, int32_t window, int32_t row, int32_t* length);

/* Size classes (multi_device.hpp): windows binned geometrically by their longest read, one BatchConfig per class, all
   classes resident and running at once. Planning is host-only. gw_poa_size_classes_run returns a gw_poa_multi handle
   (accessors above); *compute_seconds = from all workers' first fill to the last worker's end. */
typedef struct gw_poa_size_plan gw_poa_size_plan;
gw_poa_size_plan* gw_poa_plan_size_classes(int32_t n_windows, const int32_t* longest, const int32_t* reads, int32_t msa_flag,
                                           int32_t band_width, int32_t band_mode, float adaptive_storage_factor, float graph_length_factor,
                                           int32_t max_pred_distance, int32_t mismatch_score, int32_t gap_score, int32_t match_score);
void gw_poa_size_plan_destroy(gw_poa_size_plan* p);
int32_t gw_poa_size_plan_classes(gw_poa_size_plan* p);
int64_t gw_poa_size_plan_total_bytes(gw_poa_size_plan* p);
int gw_poa_size_plan_class(gw_poa_size_plan* p, int32_t k, gw_poa_batch_config* cfg, int64_t* bytes_per_window, int32_t* n_windows);
int gw_poa_size_plan_windows(gw_poa_size_plan* p, int32_t k, int32_t* window_ids);
/* keep[w] != 0: window w stays in the plan (a rank of a multi-GPU job keeps its share; configs stay those of the whole set) */
int gw_poa_size_plan_keep(gw_poa_size_plan* p, const uint8_t* keep, int32_t n_windows);
/* cudapoa::size_class_admission_gates: gates[k] = class that class k waits for on a device of `compute_units` units, -1 = none */
int gw_poa_size_plan_admission_gates(gw_poa_size_plan* p, int32_t compute_units, int32_t* gates);
gw_poa_multi* gw_poa_size_classes_run(int32_t n_windows, const int32_t* reads_per_window, const char* const* seqs, const int32_t* lengths,
                                      gw_poa_size_plan* plan, int32_t device, int64_t memory_budget, int8_t output_mask, int16_t gap_score,
                                      int16_t mismatch_score, int16_t match_score, double* compute_seconds);

/* ---- cudaaligner (aligner.hpp:76-219) ---- */
gw_aligner* gw_aligner_create_banded(int32_t max_bandwidth, void* stream, int32_t device_id, int64_t max_device_memory);
gw_aligner* gw_aligner_create(int32_t max_query_length, int32_t max_target_length, int32_t max_alignments,
                              void* stream, int32_t device_id, int64_t max_device_memory);
/* The reference's non-public aligner classes (cudaaligner/src/aligner_global_{hirschberg_myers,ukkonen,myers}.hpp), which
   its tests and benchmarks construct directly: algorithm = "default" | "hirschberg_myers" | "ukkonen" | "myers". */
gw_aligner* gw_aligner_create_algorithm(const char* algorithm, int32_t max_query_length, int32_t max_target_length,
                                        int32_t max_alignments, void* stream, int32_t device_id, int64_t max_device_memory);
void gw_aligner_destroy(gw_aligner* a);
int gw_aligner_add_alignment(gw_aligner* a, const char* query, int32_t query_length, const char* target,
                             int32_t target_length, int reverse_complement_query, int reverse_complement_target);
int gw_aligner_align_all(gw_aligner* a);
int gw_aligner_sync_alignments(gw_aligner* a);
int32_t gw_aligner_num_alignments(gw_aligner* a);
int gw_aligner_reset(gw_aligner* a);
/* Alignment accessors (alignment.hpp:37-112) */
int32_t gw_alignment_status(gw_aligner* a, int32_t i);
int32_t gw_alignment_is_optimal(gw_aligner* a, int32_t i);
int32_t gw_alignment_edit_distance(gw_aligner* a, int32_t i);
const char* gw_alignment_cigar(gw_aligner* a, int32_t i, int32_t extended, int32_t* length);
int32_t gw_alignment_states(gw_aligner* a, int32_t i, int8_t* out, int32_t cap);
/* All alignments of the last sync_alignments() at once, as run-length CIGARs in forward order: offsets[n + 1] index
   ops / counts (AlignmentState, repetitions); status[n] / optimal[n] may be NULL. Returns the total number of runs
   (call with ops == NULL to size the buffers), -1 on exception. Per-position results are run-length encoded. */
int64_t gw_aligner_get_runs(gw_aligner* a, int64_t* offsets, int8_t* ops, int32_t* counts, int64_t capacity, int32_t* status,
                            int32_t* optimal);
/* Aligner::get_alignments_device() (aligner.hpp:62-72,121): synchronises the aligner's stream, then reports the number
   of alignments and packed runs resident on the device. Returns 1 for aligners without a device-resident form. */
int gw_aligner_device_alignments(gw_aligner* a, int32_t* n_alignments, int64_t* total_length);
/* Copies the four DeviceAlignmentsPtrs arrays to host buffers (any may be NULL): cigar_operations[total],
   cigar_runlengths[total], cigar_offsets[n + 1], metadata[n]. What a device-side consumer would read in place. */
int gw_aligner_copy_device_alignments(gw_aligner* a, int8_t* cigar_operations, int32_t* cigar_runlengths, int32_t* cigar_offsets,
                                      uint32_t* metadata);
int gw_aligner_relaunch(gw_aligner* a);
/* same, timed with HIP events on the aligner's stream: all kernels of one align_all() (ms) */
int gw_aligner_relaunch_timed(gw_aligner* a, float* kernels_ms);
/* 32 * band words * target length summed over band attempts and pairs of the last align_all() (device counters) */
int gw_aligner_band_cells(gw_aligner* a, uint64_t* cells);

/* ---- cudapoa/utils.hpp: batch-shape planning (utils.cu:30-146) and window-file readers (utils.hpp:77-187) ---- */
/* Binning rule only: capacity[i] / longest[i] / reads[i] describe group i; bins_capacity may be NULL (1,2,4,... x20).
   Outputs: *n_batches, batch_cfgs[<= n_groups], groups_per_batch[<= n_groups], group_ids[n_groups] (batch by batch). */
int gw_poa_bin_groups(int32_t n_groups, const int32_t* capacity, const int32_t* longest, const int32_t* reads,
                      int32_t band_width, int32_t band_mode, float adaptive_storage_factor, float graph_length_factor,
                      int32_t max_pred_distance, const int32_t* bins_capacity, int32_t n_bins, int32_t* n_batches,
                      gw_poa_batch_config* batch_cfgs, int32_t* groups_per_batch, int32_t* group_ids);
/* get_multi_batch_sizes(): capacities from the device's free memory (needs a GPU). */
int gw_poa_get_multi_batch_sizes(int32_t n_groups, const int32_t* longest, const int32_t* reads, int32_t msa_flag,
                                 int32_t band_width, int32_t band_mode, float adaptive_storage_factor,
                                 float graph_length_factor, int32_t max_pred_distance, float gpu_memory_usage_quota,
                                 int32_t mismatch_score, int32_t gap_score, int32_t match_score, int32_t* n_batches,
                                 gw_poa_batch_config* batch_cfgs, int32_t* groups_per_batch, int32_t* group_ids);
int32_t gw_poa_estimate_max_poas(const gw_poa_batch_config* cfg, int32_t msa_flag, float gpu_memory_usage_quota,
                                 int32_t mismatch_score, int32_t gap_score, int32_t match_score);
/* Device bytes one window of this shape takes in a batch (the divisor of estimate_max_poas; host-only, no device
   query): lets a caller plan batches for a memory budget of its choice with gw_poa_bin_groups. */
int64_t gw_poa_window_device_bytes(const gw_poa_batch_config* cfg, int32_t msa_flag, int32_t mismatch_score, int32_t gap_score,
                                   int32_t match_score);
/* parse_cudapoa_file (fasta = 0, first path only) / parse_fasta_files (fasta = 1); NULL on error (gw_last_error). */
typedef struct gw_windows gw_windows;
gw_windows* gw_windows_parse(const char* const* paths, int32_t n_paths, int32_t fasta, int32_t total_windows);
void gw_windows_destroy(gw_windows* w);
int32_t gw_windows_count(const gw_windows* w);
int32_t gw_windows_num_sequences(const gw_windows* w, int32_t window);
const char* gw_windows_sequence(const gw_windows* w, int32_t window, int32_t seq, int32_t* length);

#ifdef __cplusplus
}
#endif
#endif
