// Host <-> HBM transfers of PAGEABLE host arrays (the reference's contract is NumPy in, NumPy out: SDAV.py:293-302,
// cnn_vtl.py:130-133) through the context's pinned staging ring: a small pool of host threads copies a piece between the
// caller's array and a page-locked buffer while the DMA engine moves the piece before it, so the transfer runs at the
// slower of the host's memcpy rate and the link instead of the pageable path's (a synchronous hipMemcpy of a pageable
// 429 MB array: 40-60 ms; staged: ~12 ms).  The ring and the threads are created on first use and live with the context.
#include "dlc_internal.h"

#include "host_staging_impl.h"
