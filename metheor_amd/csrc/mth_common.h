// mth_common.h -- shared definitions of the gfx950 engine (host + device).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/metheor_hip.h"

namespace mth {

// ---- tiling constants -------------------------------------------------------------------
// A tile is TILE_W consecutive reference positions of one contig; one workgroup owns the sites
// of one tile and keeps their accumulators in LDS.  The read index has one entry per IDX_Q bp.
// (the tile width is a template parameter of the tile kernel: 1024 / 2048 / 4096)
constexpr int IDX_QSHIFT = 5;   // 32-bp quanta (256 until round 2: up to 362 bp of useless candidates per tile / site)
constexpr int IDX_Q = 1 << IDX_QSHIFT;
constexpr int BLOCK = 256;

// pdr.rs:162 -- a site is flushed once a passing read's first CpG lies more than 150 bp past it
constexpr int PDR_FLUSH_MARGIN = 150;

// device-side error bits (DevState::err)
enum : uint32_t {
    ERRB_UNSORTED = 1u << 0,
    ERRB_SPAN = 1u << 1,
    ERRB_RANGE = 1u << 2,
    ERRB_CAPACITY = 1u << 3,
    ERRB_FORMAT = 1u << 5,     // device record decode: malformed BAM record
    ERRB_NOXM = 1u << 6,       // device record decode: record without XM:Z
    ERRB_CRC = 1u << 8,        // device inflate: CRC32 of an inflated BGZF block does not match its trailer
    ERRB_TAGPANIC = 1u << 9,   // tag: a record on which the reference's determine_xm_tag_string panics (tag.rs:24, 155-170, 297)
    ERRB_FDRPPANIC = 1u << 10, // FDRP / qFDRP: a read on which the reference's window index leaves 0..=402 (fdrp.rs:70-72) -- it panics
    ERRB_UNALIGNED = 1u << 7,  // device record walk: a record straddles two BGZF blocks (take the host walk)
};

// small block of device-resident state, read back by the synchronising getters
struct DevState {
    uint32_t err;
    uint32_t n_batches;
    uint64_t n_sites;        // PDR rows emitted so far (all batches)
    uint64_t cur_base;       // n_sites before the batch in flight
    int64_t  lpmd[4];        // n_concordant, n_discordant, n_read, n_valid_read
    uint32_t safe_hi;        // batch in flight: tiles whose candidate reads end at or before this read index can load 8 call slots
                             // per read without running past the call arrays (k_build_index; spares the tile a dependent load)
    uint32_t pad_;           // 64 bytes: the runtime clears a 16-byte multiple with ONE fill kernel (56 bytes took two)
};
static_assert(sizeof(DevState) == 64, "DevState is cleared with one aligned fill");

struct SiteRec {  // per-tile scratch row
    int32_t  pos;
    uint32_t n_conc, n_disc, pad;
};

enum KernelId { K_INDEX = 0, K_TILE, K_GATHER, K_QBOUND, K_QTILE, K_QINSERT, K_QEMIT, K_MHLWALK, K_MHLWALKBIG, K_MHLEMIT, K_PDRWALK, K_FDRPWALK, K_FDRPEMIT, K_PAIRS, K_PAIRSTILE, K_DECODE, K_INFLATE, K_CRC, K_MHLTILE, K_WIDE, K_FDRPTILE, K_FDRPCHAIN, K_FDRPWALK4, K_FDRPWTILE, K_MHLROWCHK, K_NUM };

}  // namespace mth
