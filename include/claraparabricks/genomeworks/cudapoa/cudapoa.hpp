// cudapoa.hpp -- status codes, band modes and output types of the POA module.
// Source-compatible with the reference's cudapoa/cudapoa.hpp:34-85; enumerator VALUES are part of the
// device <-> host protocol (a failing window stores its StatusType in consensus[1]).
#pragma once
#include <string>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudapoa
{

enum StatusType
{
    success = 0,
    exceeded_maximum_poas,
    exceeded_maximum_sequence_size,
    exceeded_maximum_sequences_per_poa,
    node_count_exceeded_maximum_graph_size,
    edge_count_exceeded_maximum_graph_size,
    exceeded_adaptive_banded_matrix_size,
    exceeded_maximum_predecessor_distance,
    loop_count_exceeded_upper_bound,
    output_type_unavailable,
    zero_weighted_poa_sequence,
    empty_poa_group,
    generic_error
};

/// Human-readable message + hint for a status code.
void decode_error(StatusType error_type, std::string& error_message, std::string& error_hint);

/// How much of the score matrix is computed per read (see the reference header for the trade-offs).
enum BandMode
{
    full_band = 0,
    static_band,
    adaptive_band,
    static_band_traceback,
    adaptive_band_traceback
};

/// Initialise the module (logging at WARN).
StatusType Init();

enum OutputType
{
    consensus = 0x1,
    msa       = 0x1 << 1
};

} // namespace cudapoa
} // namespace genomeworks
} // namespace claraparabricks
