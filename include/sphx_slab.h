/*
 * sphx_slab.h — C ABI of the native multi-GPU layer: SPHSystem::step() over x-slabs of the domain, one
 * process per GPU, halo exchange with RCCL point-to-point messages over xGMI (libsphx.so, csrc/slab.hip).
 *
 * Why x-slabs: the reference's linear cell id runs x slowest (CUDAFunctions.cuh:68), so the particles of the cell
 * columns [x0, x1) plus one ghost column per side are ONE contiguous range of the globally cell-sorted arrays.
 * Every slab holds [left ghosts | owned | right ghosts], entry for entry a slice of the single-device array, so
 * each per-particle sum visits the same neighbours in the same order: the distributed run is bit-identical to
 * SPHSystem::step() on one device for any number of slabs (tests/test_gpu_slab.py).
 *
 * Every slab's engine works on the whole grid and the whole boundary set and is only handed its own particles, so
 * the cut planes are two numbers of the driver: they are re-balanced while the fluid spreads (sphx_slab_set_rebalance).
 *
 * One step = particle exchange (migrants and ghost copies, after last step's advect) -> local cell sort -> the
 * solver's stages (sphx_phase in sphx_c.h) with a halo refresh after every stage that writes a field the next
 * stage reads from neighbours.  With overlap on (default) a stage runs on the two edge layers first, the halo
 * messages of its output start on a separate stream, and the interior particles are swept meanwhile.
 *
 * Transports: RCCL (one process per GPU; ncclSend/ncclRecv grouped per exchange, ncclAllReduce of one integer per
 * iteration in adaptive DFSPH) and loopback (all slabs in this process on the current device; device-to-device
 * copies) for single-GPU testing.
 */
#ifndef SPHX_SLAB_H
#define SPHX_SLAB_H

#include "sphx_c.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sphx_slab_group sphx_slab_group;   /* the slabs driven by this process: 1 or a few (RCCL), or all (loopback) */

enum {
    SPHX_SLAB_NO_OVERLAP = 1,       /* run every stage on all particles, then exchange (the simple schedule) */
    SPHX_SLAB_SWEEP_GHOSTS = 2      /* with NO_OVERLAP: also sweep the ghost particles (what the r01 Python driver did) */
};

/* 128-byte RCCL bootstrap token: rank 0 creates it, the launcher hands it to every rank (any side channel) */
int sphx_slab_rccl_unique_id(char id128[128]);

/*
 * Cuts the GLOBAL scene (the same arrays on every rank; fluid velocities optional) into `world` x-slabs balancing
 * the initial particle counts and creates the slabs [first_rank, first_rank + local_ranks).
 *   RCCL:     rccl_id128 = the token, one call per process after sphx_set_device; normally local_ranks = 1 (one slab
 *             per GPU).  A process may drive local_ranks > 1 CONSECUTIVE slabs (the same number in every process,
 *             first_rank a multiple of it): the communicator then has world / local_ranks ranks and slabs of one
 *             process talk through RCCL sends to self.
 *   loopback: first_rank = 0, local_ranks = world, rccl_id128 = NULL
 * params: the whole-domain scalars (as for sphx_create); DFSPH runs fixed or adaptive iterations as params say.
 */
int sphx_slab_create(const sphx_params *params, const float *fluid_xyz, const float *fluid_vel_or_null, int n_fluid,
                     const float *boundary_xyz, int n_boundary, int world, int first_rank, int local_ranks,
                     const char *rccl_id128, int flags, sphx_slab_group **out);
int sphx_slab_destroy(sphx_slab_group *g);

/* n steps; the k-th step since creation equals the k-th SPHSystem::step() of the single-device system counting its
 * constructor step.  *ms_total: wall time of the batch on this process (host clock, device synchronised).        */
int sphx_slab_step(sphx_slab_group *g, int n, float *ms_total);

/* Cut re-balancing: every `every_steps` steps (<= 0: never; default 16) a cut plane moves by one cell column
 * towards the lighter of its two slabs when their owned particle counts differ by more than `tolerance`
 * (relative; default 0.05).  The dam break spreads along x, so static cuts lose balance exactly when the
 * interesting part starts.  Must be set identically on every rank.  Results do not depend on it.       */
int sphx_slab_set_rebalance(sphx_slab_group *g, int every_steps, float tolerance);

/* Host-only helpers (no GPU needed): the initial decomposition sphx_slab_create would choose — cuts[world + 1] with
 * cuts[0] = 0, cuts[world] = params->cells[0], optional counts[world] of fluid particles per slab — and the
 * re-balancing rule both neighbours of a cut evaluate: -1 = the left slab hands its last column to the right one,
 * +1 = the opposite, 0 = stay.                                                                          */
int sphx_slab_plan_cuts(const sphx_params *params, const float *fluid_xyz, int n_fluid, int world, int *cuts, long long *counts);
/* ... and the particle capacity each slab's engine would be created with (the same on every rank): twice the most
 * particles a slab holds at the start (owned + ghost columns + two columns of head-room for a moving cut), at most the
 * whole scene.  A slab that outgrows it makes sphx_slab_step fail with SPHX_ERR_STATE.                       */
int sphx_slab_plan_capacity(const sphx_params *params, const float *fluid_xyz, int n_fluid, int world, long long *capacity);
int sphx_slab_cut_rule(long long owned_left, long long owned_right, int width_left, int width_right, int ghost, float tolerance);

/* geometry and sizes of local slab `index`: owned cell columns [x0, x1), particles owned / held incl. ghosts */
int sphx_slab_info(const sphx_slab_group *g, int index, int *x0, int *x1, int *owned, int *held);
/* owned particles of local slab `index` to host arrays of `capacity` particles (ids: original indices) */
int sphx_slab_gather(sphx_slab_group *g, int index, int capacity, int *ids, float *pos_xyz, float *vel_xyz,
                     float *density, int *count);
/* iteration counts of the last DFSPH step (identical on every rank) */
int sphx_slab_iters(const sphx_slab_group *g, int *divergence_iters, int *density_iters);
/* the engine system of local slab `index` (for sphx_kernel_timer / sphx_device_ptr); owned by the group.
 * A slab's system sorts inside the window of cell columns it holds: SPHX_F_CELLSTART_F read from it is defined for the cells of
 * the held columns (ghost columns included) only; cells outside that window keep whatever an earlier step left there.        */
int sphx_slab_system(const sphx_slab_group *g, int index, sphx_system **sys);
/* The transport as it reports itself (bench.py --gpus N puts this into its line so that a scaling run can be checked):
 * transport_kind 0 = loopback copies, 1 = RCCL; comm_ranks / comm_rank = ncclCommCount / ncclCommUserRank of the
 * communicator THIS process joined (-1: the library does not export them); counters4 = payload bytes sent, payload bytes
 * received, grouped send/recv rounds and all-reduces posted by this process since creation.                             */
int sphx_slab_comm_info(const sphx_slab_group *g, int *transport_kind, int *comm_ranks, int *comm_rank, long long counters4[4]);
/* seconds this process spent in host-side waits for exchanges since creation (diagnostic) */
int sphx_slab_wait_seconds(const sphx_slab_group *g, double *seconds);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_SLAB_H */
