// cudaaligner.hpp -- enums of the pairwise alignment module (source-compatible with the reference's
// cudaaligner/cudaaligner.hpp:34-68; AlignmentState values are part of the device result encoding).
#pragma once
#include <cstdint>

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

enum StatusType
{
    success = 0,
    uninitialized,
    exceeded_max_alignments,
    exceeded_max_length,
    exceeded_max_alignment_difference,
    generic_error
};

enum AlignmentType
{
    global_alignment = 0,
    unset
};

/// One position of an alignment.
enum AlignmentState : int8_t
{
    match = 0,
    mismatch,
    insertion, ///< absent in query, present in target
    deletion   ///< present in query, absent in target
};

enum CigarFormat
{
    basic = 0, ///< symbols M, I, D
    extended   ///< symbols =, X, I, D
};

/// Initialise the module (logging at WARN).
StatusType Init();

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
