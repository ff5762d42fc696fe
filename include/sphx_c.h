/*
 * sphx_c.h — C ABI of the MI355X-native SPH engine (libsphx.so).
 *
 * The reference (zhai-xiao/CPP-Fluid-Particles) exposes its hot path as C++ classes, not as a C
 * plugin interface; the source-compatible C++ mirror of those classes lives next to this file
 * (DArray.h, Particles.h, SPHParticles.h, BaseSolver.h, BasicSPHSolver.h, DFSPHSolver.h,
 * PBDSolver.h, SPHSystem.h).  This header is the flat `extern "C"` boundary a binding (ctypes,
 * cgo, JNI, ...) uses: plain pointers and sizes, no C++ or torch types.  Every entry point names
 * the reference interface it stands for.  INTEGRATION.md shows the reference-side stub.
 *
 * All functions return 0 on success or a negative sphx_status; sphx_last_error() gives text.
 * There is no CPU fallback: without a HIP device sphx_create fails with SPHX_ERR_NO_DEVICE.
 */
#ifndef SPHX_C_H
#define SPHX_C_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sphx_status {
    SPHX_OK = 0,
    SPHX_ERR_INVALID = -1,     /* bad argument / size mismatch */
    SPHX_ERR_NO_DEVICE = -2,   /* no HIP device visible */
    SPHX_ERR_HIP = -3,         /* a HIP runtime call failed */
    SPHX_ERR_STATE = -4        /* call not valid in the current state */
} sphx_status;

enum { SPHX_ARITH_STRICT = 0, SPHX_ARITH_TOLERANCE = 1, SPHX_ARITH_TOLERANCE_PERSISTENT = 2 };
/* solver kinds: the three BaseSolver implementations selected in main.cpp:119-130 */
enum { SPHX_WCSPH = 0, SPHX_DFSPH = 1, SPHX_PBD = 2 };

/*
 * Scalars of one simulation: the SPHSystem constructor arguments (SPHSystem.h:22-38) — which are
 * also the 14 scalars of BaseSolver::step (BaseSolver.h:22-26) plus m0 — followed by the solver
 * constructor knobs (DFSPHSolver.h:27-39, PBDSolver.h:27-38).
 */
typedef struct sphx_params {
    float space[3];            /* spaceSize */
    int   cells[3];            /* cellSize = ceil(spaceSize / cellLength), main.cpp:67 */
    float cell_length;         /* sphCellLength */
    float radius;              /* sphSmoothingRadius */
    float dt;
    float m0;                  /* sphM0, uniform fluid particle mass */
    float rho0;                /* sphRho0 */
    float rho_boundary;        /* sphRhoBoundary */
    float stiff;               /* sphStiff (WCSPH Tait EOS) */
    float visc;                /* sphVisc */
    float surface_tension;     /* sphSurfaceTensionIntensity */
    float air_pressure;        /* sphAirPressure */
    float gravity[3];          /* sphG */
    int   solver;              /* SPHX_WCSPH / SPHX_DFSPH / SPHX_PBD */
    float dfsph_density_thr;   /* DFSPHSolver ctor: defaultDensityErrorThreshold (1e-3) */
    float dfsph_divergence_thr;/* DFSPHSolver ctor: defaultDivergenceErrorThreshold (1e-3) */
    int   dfsph_max_iter;      /* DFSPHSolver ctor: defaultMaxIter (20) */
    int   dfsph_fixed_div;     /* <0: adaptive loop of DFSPHSolver.cu:347-361; >=0: exactly that many */
    int   dfsph_fixed_den;     /* <0: adaptive loop of DFSPHSolver.cu:187-208; >=0: exactly that many */
    int   pbd_iters;           /* PBDSolver ctor: defaultMaxIter (20), always all are run */
    float pbd_xsph_c;          /* PBDSolver ctor: defaultXSPH_c (0.05) */
    float pbd_relaxation;      /* PBDSolver ctor: defaultRelaxation (0.75) */
    int   pow7_mode;           /* must be 0 (fp64 multiply chain for the Tait exponent) */
    int   xsph_mode;           /* must be 0 (Jacobi XSPH) */
    int   reserved[4];         /* reserved[0]: engine switches for tests (bit0 unfused schedule,
                                  bit1 direct 27-cell walks instead of the neighbour list,
                                  bit2 LDS-staged 64-particle tiles);
                                  reserved[1], reserved[2]: slab sub-grid (see sphx_slab.h);
                                  reserved[3]: arithmetic of the neighbour sweeps, SPHX_ARITH_STRICT (0, default:
                                  every bit equals the IEEE evaluation of the reference's expressions) or
                                  SPHX_ARITH_TOLERANCE (1: hardware rsq/rcp + fused multiply-adds, ~1e-7 per pair term;
                                  the reference's own binary is built -use_fast_math, src/CMakeLists.txt:43) or
                                  SPHX_ARITH_TOLERANCE_PERSISTENT (2: the same arithmetic, and the neighbour rows of WCSPH /
                                  DFSPH carry a skin and survive from step to step until a device-side check finds that
                                  some particle has moved more than 0.49 skin relative to the others; the API arrays, cell
                                  indices and the cell table are still brought up to date in the reference's order by
                                  every step -- SPHSystem::setPersistentRows in SPHSystem.h; needs cell_length > radius,
                                  whole-domain systems only; PBD runs as 1) */
} sphx_params;

/* device-resident fields readable through sphx_get (host copy) / sphx_device_ptr (raw pointer) */
typedef enum sphx_field {
    SPHX_F_POS = 0,        /* float[3n]  SPHParticles::getPosPtr,  Particles.h:32-34   */
    SPHX_F_VEL,            /* float[3n]  SPHParticles::getVelPtr,  Particles.h:35-37   */
    SPHX_F_DENSITY,        /* float[n]   getDensityPtr, SPHParticles.h:40-42          */
    SPHX_F_PRESSURE,       /* float[n]   getPressurePtr, SPHParticles.h:34-36         */
    SPHX_F_MASS,           /* float[n]   getMassPtr, SPHParticles.h:49-51             */
    SPHX_F_CELL,           /* int[n]     getParticle2Cell (pre-sort order), :46-48    */
    SPHX_F_CELLSTART_F,    /* int[C+1]   SPHSystem::cellStartFluid, SPHSystem.h:67    */
    SPHX_F_CELLSTART_B,    /* int[C+1]   SPHSystem::cellStartBoundary, SPHSystem.h:68 */
    SPHX_F_ID,             /* int[n]     original index of the particle now in slot q  */
    SPHX_F_BPOS,           /* float[3nb] boundary positions (cell-sorted)              */
    SPHX_F_BMASS,          /* float[nb]  boundary masses, SPHSystem.cu:79-112          */
    SPHX_F_ALPHA,          /* float[n]   DFSPHSolver::alpha                            */
    SPHX_F_KAPPA,          /* float[n]   DFSPHSolver::bufferFloat (stiffness)          */
    SPHX_F_ERROR,          /* float[n]   DFSPHSolver::error                            */
    SPHX_F_WARM,           /* float[n]   DFSPHSolver::denWarmStiff                     */
    SPHX_F_POS_LAST,       /* float[3n]  PBDSolver::fluidPosLast                       */
    SPHX_F_LAMBDA,         /* float[n]   PBDSolver::bufferFloat (lambda)               */
    SPHX_F_BUF3,           /* float[3n]  BasicSPHSolver::bufferFloat3 (colour gradient)*/
    SPHX_F_VEL4,           /* float[4n]  engine mirror of vel (x,y,z,0), what sweeps gather    */
    SPHX_F_CG4,            /* float[4n]  engine mirror of the colour gradient                  */
    SPHX_F_PTERM,          /* float[n]   p / max(EPS, rho^2), the neighbour term of the pressure force */
    SPHX_F_POS4,           /* float[4n]  engine mirror of pos (x,y,z,mass), what sweeps gather (PBD halo target) */
    SPHX_F_POSF,           /* float[4n]  (x,y,z, scalar neighbour field: kappa / pressure term / lambda): what the one-gather
                              sweeps read; halo target next to the scalar's own array */
    SPHX_F_COUNT_
} sphx_field;

typedef struct sphx_system sphx_system;   /* opaque: owns SPHParticles x2, a solver, an SPHSystem */

/*
 * Engine tuning: the behaviour switches of the engine that are not part of the simulation (schedules, row capacity, which
 * sweeps run quad-per-particle, loop windows, stream priorities of the slab layer).  Until r04 these were SPHX_* environment
 * variables read inside the library; the library now reads NO environment variable for its behaviour (SPHX_RCCL_LIBRARY, the
 * path of the RCCL build to open, is the one exception).  Process-wide: sphx_tuning_defaults fills the defaults,
 * sphx_set_tuning installs a block; systems and slab groups read it when they are CREATED, the fields marked (live) at every
 * step.  -1 (or 0 where noted) = the engine's own default.  Results never depend on these switches (every combination is
 * bit-exact in strict arithmetic and inside the contract in tolerance arithmetic: tests/test_gpu_parity.py).
 */
typedef struct sphx_tuning {
    int   struct_size;        /* sizeof(sphx_tuning) of the caller: a binding built against another layout is refused */
    int   engine_flags;       /* OR-ed into sphx_params.reserved[0] (bit0 unfused, bit1 no rows, bit2 LDS tiles, bit3 linear tiles, bit4 no quads) */
    int   row_capacity;       /* entries per neighbour row: 0 = adaptive from 48 (slabs: 96 fixed), 8..1024 = fixed */
    int   quad_mask;          /* strict arithmetic: sweeps that walk rows quad-per-particle (-1: the rate sweeps) */
    int   duo_mask;           /* ... with two lanes per particle (-1: head and viscosity+colour from 4 M particles on) */
    int   quad_mask_tol;      /* tolerance arithmetic: quad walks (-1: 15, of which the corrections only from 4 M particles on) */
    int   tol_strict_rate;    /* tolerance arithmetic, >= 4 M particles: rate sweeps on the strict quad kernel (-1: no since r05) */
    int   brick;              /* 1: the opt-in compact-brick LDS stage of tolerance arithmetic (measured slower; 0) */
    int   brick_min;          /* particles from which the brick stage is used (0: 2,000,000) */
    int   range_order;        /* range-restricted launches of wide slabs keep the (y-chunk, x) tile schedule (-1: yes) */
    int   range_order_min;    /* ... from this many particles per range on (0: 3,000,000) */
    int   force_tile_order;   /* 1: build the tile schedule on small grids too (tests) */
    int   no_fastmath;        /* 1: plain IEEE operators instead of the validated exact fast sqrt / division paths */
    int   no_graph;           /* (live) 1: sphx_step_n launches eagerly instead of replaying a captured hipGraph */
    int   graph_debug;        /* (live) 1: say on stdout why a capture failed */
    int   dfsph_host_loop;    /* 1: adaptive DFSPH decides its loops on the host (one read-back per iteration), as the reference does */
    int   dfsph_window;       /* (live) iterations of a device-decided loop enqueued as ordinary launches: -1 follow the counts, >= 0 fixed */
    int   dfsph_no_tail;      /* (live) 1: no persistent loop-tail launch, gated launches for every possible iteration */
    int   no_kick_fusion;     /* (live) 1: the gravity kick of fixed-count DFSPH stays a pass of its own */
    float pbd_skin;           /* PBD skin rows: skin as a fraction of the radius (< 0: 0.05; 0: one row build per Jacobi iteration) */
    int   pbd_skin_fixed;     /* 1: no controller that drops the skin in violent phases */
    int   persist_controller; /* persistent rows: leave the mode for 256 steps while (nearly) every step rebuilds its rows (-1: yes) */
    int   slab_edge_stream;   /* slab layer: edge layers of a DFSPH / WCSPH stage on a stream of their own beside the interior (-1: yes) */
    int   slab_comm_priority; /* RCCL transport's communication stream: 0 highest priority (default), 1 default priority, 2 lowest */
    int   dfsph_tail_flat;    /* (live) 1: the loop tail's sweeps are separated by the r04 barrier (one counter, a fence pair per block) instead of the XCD-hierarchical one */
    int   group_build_max;    /* particles up to which the row builder works with 16 lanes per particle (0: 81,920; < 0: never) */
    int   pbd_no_partial;     /* 1: PBD skin rows are rebuilt as a whole when a particle changes its cell inside a step (until r05) instead of row by row */
    int   reserved[5];
} sphx_tuning;
int  sphx_tuning_defaults(sphx_tuning *out);
int  sphx_set_tuning(const sphx_tuning *tuning);       /* NULL: back to the defaults */
int  sphx_get_tuning(sphx_tuning *out);

/* library / device ------------------------------------------------------------------------- */
const char *sphx_last_error(void);
int  sphx_device_count(void);                         /* hipGetDeviceCount; 0 when no GPU */
int  sphx_set_device(int ordinal);                    /* hipSetDevice (one process per GPU) */
int  sphx_sizeof_params(void);
/* PCI bus id of HIP device `ordinal` ("0000:05:00.0"): lets a multi-process run state which physical device each rank used */
int  sphx_device_pci_id(int ordinal, char *out, int capacity);

/* scene of main.cpp:54-117, scaled by nx/24 as BASELINE.md §4 prescribes (nx=24: the reference
 * scene).  Two-call protocol: counts first, then fill caller-owned host buffers.              */
int  sphx_scene_params(int nx, sphx_params *out);
int  sphx_scene_counts(int nx, int *n_fluid, int *n_boundary);
int  sphx_scene_fill(int nx, float *fluid_xyz, float *boundary_xyz);

/* SPHSystem::SPHSystem (SPHSystem.cu:33-77) incl. make_shared<SPHParticles> x2 and the solver
 * (main.cpp:86,117,119-134).  Host buffers are copied (Particles.h:22-25).  run_ctor_step=1
 * reproduces the reference constructor, which ends with one full step() (SPHSystem.cu:76).   */
int  sphx_create(const sphx_params *params, const float *fluid_xyz, int n_fluid,
                 const float *boundary_xyz, int n_boundary, int run_ctor_step, sphx_system **out);
int  sphx_destroy(sphx_system *sys);

/* SPHSystem::step (SPHSystem.cu:129-158): neighbour search + solver step + device sync;
 * *ms receives the hipEvent-timed duration the reference returns.                             */
int  sphx_step(sphx_system *sys, float *ms);
/* n back-to-back steps with one sync at the end (fixed-iteration modes replay a captured
 * hipGraph); *ms_total is the hipEvent time of the whole batch.                               */
int  sphx_step_n(sphx_system *sys, int n, float *ms_total);

/* Stage-wise DFSPH step for distributed drivers (x-slab decomposition).  The
 * stages are the fused schedule of DFSPHSolver::step (DFSPHSolver.cu:33-72 restated); between two
 * stages the driver refreshes the halo copy of the field the next stage reads from neighbours.
 * A slab system is created with run_ctor_step = 0, params.cells[0] = local cell columns (owned +
 * one ghost layer per side) and params.reserved[1] = global x index of local column 0.          */
typedef enum sphx_phase {
    SPHX_PH_SEARCH = 0,       /* neighbour search, pack, neighbour rows, warm stiffness follows sort */
    SPHX_PH_HEAD,             /* density, alpha, first divergence error -> error, kappa            */
    SPHX_PH_DIV_CORRECT,      /* vel += sum m (k_i + k_j) gradW                 (writes vel)       */
    SPHX_PH_DIV_ERROR,        /* divergence error                               (writes kappa)     */
    SPHX_PH_FORCE,            /* gravity                                                            */
    SPHX_PH_VISC_COLOR,       /* viscosity delta-v + colour gradient            (writes cg)        */
    SPHX_PH_SURFACE,          /* vel += deltaV, surface tension + air pressure  (writes vel)       */
    SPHX_PH_WARM_CORRECT,     /* warm start: density correction with last step's stiffness (vel)   */
    SPHX_PH_DEN_ERROR_SET,    /* density error, warm = kappa                    (writes kappa)     */
    SPHX_PH_DEN_CORRECT,      /* density correction                             (writes vel)       */
    SPHX_PH_DEN_ERROR_ACC,    /* density error, warm += kappa                   (writes kappa)     */
    SPHX_PH_ADVECT,           /* pos += dt vel, box clamp                                           */
    /* WCSPH stages (BasicSPHSolver, fused schedule of BasicSPHSolver.cu:237-260) */
    SPHX_PH_W_SEARCH,         /* neighbour search, pack + gravity kick, neighbour rows              */
    SPHX_PH_W_PROPS,          /* viscosity delta-v, colour gradient, density, pressure (writes cg, pterm) */
    SPHX_PH_W_SURFACE,        /* vel += deltaV, surface tension + air pressure                      */
    SPHX_PH_W_PRESSURE,       /* pressure force                                                     */
    /* PBD schedule (PBDSolver.cu:34-79): P_SEARCH, pbd_iters x [P_LAMBDA, P_DELTA], P_VELOCITY, P_XSPH,
     * P_SURFACE, P_TAIL.  PBD sweeps run on positions that moved after binning, so a slab keeps TWO
     * ghost columns per side; refresh lambda after P_LAMBDA, the position mirror (SPHX_F_POS4) after
     * P_DELTA, the velocity mirror after P_VELOCITY and the colour gradient mirror after P_XSPH. */
    SPHX_PH_P_SEARCH,         /* neighbour search, last positions follow the sort, pack             */
    SPHX_PH_P_LAMBDA,         /* neighbour rows for the current positions, density + lambda (writes lambda) */
    SPHX_PH_P_DELTA,          /* delta-p sweep, pos += delta-p, box clamp       (writes pos mirror) */
    SPHX_PH_P_VELOCITY,       /* vel = (pos - pos_last) / dt                    (writes vel mirror) */
    SPHX_PH_P_XSPH,           /* XSPH viscosity (+ colour gradient)             (writes cg)        */
    SPHX_PH_P_SURFACE,        /* surface tension + air pressure                                     */
    SPHX_PH_P_TAIL,           /* gravity, remember positions, predict (advect + clamp)              */
    /* fused forms (one row walk, same bits): */
    SPHX_PH_SURFACE_WARM,     /* DFSPH: SPHX_PH_SURFACE + SPHX_PH_WARM_CORRECT (surface effects on; writes vel)     */
    SPHX_PH_W_SURFACE_PRESSURE,/* WCSPH: SPHX_PH_W_SURFACE + SPHX_PH_W_PRESSURE (surface effects on)               */
    /* PBD, split form for range-restricted launches (slab layer: interior first, edges after the halo arrived):
     * P_DELTA = P_DELTA_SWEEP on every range, THEN P_APPLY (Jacobi: every delta-p is computed from the old positions) */
    SPHX_PH_P_DELTA_SWEEP,    /* delta-p sweep only                              (writes the delta-p buffer)       */
    SPHX_PH_P_APPLY           /* pos += delta-p, box clamp                       (writes pos mirror)               */
} sphx_phase;
int  sphx_run_phase(sphx_system *sys, int phase);
/* adaptive DFSPH across processes: the error stages (DIV_ERROR, DEN_ERROR_ACC) accumulate the exact
 * 2^-32 fixed-point |error| sum over particles [lo, hi) only (a slab's owned range) when reduce != 0;
 * sphx_error_total_fixed returns that integer so the driver can all-reduce it (order-independent). */
int  sphx_run_phase_reduce(sphx_system *sys, int phase, int lo, int hi);
int  sphx_error_total_fixed(sphx_system *sys, long long *total);
/* number of fluid particles in use (<= the n_fluid capacity given to sphx_create); slab drivers
 * change it every step as particles migrate between processes                                   */
int  sphx_set_count(sphx_system *sys, int n_fluid);
/* make the engine enqueue on a caller-owned hipStream_t (e.g. torch's current stream) so that the
 * caller's collectives are ordered with the engine's kernels; call before sphx_create.          */
int  sphx_use_stream(void *hip_stream);
int  sphx_sync(void);                                  /* hipStreamSynchronize(engine stream) */
/* global cell column (int)(x / cell_length) of n DEVICE positions (xyz triples), computed with the
 * engine's own division so that a driver agrees with the grid about column membership          */
int  sphx_cell_columns(const float *device_xyz, int n, float cell_length, int *device_out);

/* SPHSystem::size/boundarySize (SPHSystem.h:44-55) and grid size */
int  sphx_counts(const sphx_system *sys, int *n_fluid, int *n_boundary, int *n_cells);
/* the scalars the system was created with (e.g. after sphx_snapshot_load) */
int  sphx_get_params(const sphx_system *sys, sphx_params *out);
/* neighbour statistics of the most recent row build (bench.py reports them beside the timings): total
 * accepted pairs, longest row, histogram of row lengths (bin 127 = 127 and more; may be NULL)   */
int  sphx_row_stats(const sphx_system *sys, long long *total_pairs, int *longest_row, int *hist128);
/* what ragged rows cost the quad-per-particle walk (a wave runs to the longest of its 16 rows), from the row lengths of the last build:
 * out6 = {waves, chunk steps as walked, chunk steps if all rows were even, chunk steps with every walk cut at `cut` entries,
 * chunk steps of a compact second launch over the tails beyond `cut`, particles longer than `cut`} (bench.py: post-impact legs) */
int  sphx_row_walk_stats(const sphx_system *sys, int cut, long long out6[6]);
/* current capacity (entries per particle) of the neighbour rows: 96 fixed for slabs / SPHX_NBR_CAP, else adaptive from 48 */
int  sphx_row_capacity(const sphx_system *sys, int *capacity);
/* PBD diagnostics: how many times since creation the once-per-step neighbour rows had to be rebuilt inside a step because
 * a particle moved farther than their skin allows (decided and done on the device; always 0 for other solvers)      */
int  sphx_rows_stale(const sphx_system *sys, int *rebuilds);
/* PBD skin rows: launches so far that rebuilt only the rows of particles which had changed their cell (r06; 0 for the other solvers) */
int  sphx_rows_partial(const sphx_system *sys, int *partial_rebuilds);
/* persistent rows (reserved[3] = 2): 1/0 whether the mode is in use for this system, row builds and steps since creation
 * (both 0 when it is not in use: every step builds its rows then)                                                      */
int  sphx_persistent_stats(const sphx_system *sys, int *in_use, int *row_builds, int *steps);
/* iteration counts of the last DFSPH step (the values DFSPHSolver.cu:49,65 compute and drop) */
int  sphx_iters(const sphx_system *sys, int *divergence_iters, int *density_iters);

/* field access: blocking D2H / H2D copies of whole fields, and raw device pointers */
int  sphx_field_bytes(const sphx_system *sys, int field, size_t *bytes);
int  sphx_get(const sphx_system *sys, int field, void *host_dst, size_t bytes);
int  sphx_set(sphx_system *sys, int field, const void *host_src, size_t bytes);   /* POS, VEL, WARM, BMASS, POS_LAST, ID */
/* Raw device pointer.  With persistent rows (reserved[3] = 2) the solver steps a working copy and the API fields (POS ... BMASS)
 * are EXPORTED by every step: read them freely, but after WRITING one of them through the pointer call sphx_invalidate_order()
 * before the next step, or the write is overwritten by the next export (sphx_set does this itself).  Asking for the pointer of a
 * solver-internal field (ALPHA ... POSF) flushes the mode (arrays permuted into API order, rows rebuilt by the next step);
 * sphx_get reads such fields through the slot map without disturbing it.                                                      */
int  sphx_device_ptr(const sphx_system *sys, int field, void **device_ptr);
/* tell a persistent-rows system that its API arrays were written in place: the next step re-primes its working copy from them
 * and rebuilds the rows (no-op for other systems; SPHSystem::invalidatePersistentOrder)                                        */
int  sphx_invalidate_order(sphx_system *sys);

/* per-kernel timing of the last sphx_profile_step: names/ms arrays of up to cap entries */
int  sphx_profile_step(sphx_system *sys, int cap, char (*names)[48], float *ms, int *count);

/* live per-kernel timing (bench.py roofline leg): hipEvents on the engine stream around each
 * launch whose span name equals `filter` (NULL/"" = all).  While enabled sphx_step_n launches
 * eagerly.  collect() synchronises and returns, per span name, total ms and launch count.       */
int  sphx_kernel_timer(int enable, const char *filter);
int  sphx_kernel_timer_collect(int cap, char (*names)[48], float *total_ms, int *launches, int *count);
/* name (>= 48 bytes) of the kernel instantiation the most recent DFSPH error sweep was launched as; returns its variant number
 * (>= 0): 1 strict quad walk, 2 tolerance quad walk, 3 strict quad walk serving a tolerance-mode step, 4 duo, 5 lane, 6 LDS tiles, 7 brick */
int  sphx_last_rate_kernel(char *name, int capacity);

/* pointwise evaluation of the four smoothing kernels of CUDAFunctions.cuh:23-98 on the device
 * (device-function parity test): r3 = n displacement vectors, outputs W[n], gradW[3n],
 * viscosity laplacian[n], surface-tension gradient[3n].  Host pointers.                       */
int  sphx_eval_kernels(const float *r3, int n, float radius, float *W, float *gradW,
                       float *visc_lap, float *surf_grad);

/* IEEE self-test: evaluates a/b, sqrt(a), (int)(a/b), a*b+c (uncontracted) on the device for n
 * host-supplied operands so a test can compare them bit-for-bit with the host.               */
int  sphx_ieee_probe(const float *a, const float *b, const float *c, int n,
                     float *quot, float *root, int *trunc, float *muladd);

/* Self-test of the engine's exact fast paths for sqrt and division (sph_device.hpp): counts
 * bit mismatches against the plain IEEE operators for {x/R over every float x in [0, 2.2R]; sqrt
 * over every non-negative finite float; shared-denominator division over `samples` pseudo-random
 * operand triples}.  enabled2 = {fastQ, fastDiv} as the engine would set them for this radius.   */
int  sphx_fastmath_selftest(float radius, unsigned long long samples, unsigned int *mismatches3, int *enabled2);

/* generate_dots (vbo.cu:26-51): position copy + density colour ramp into caller device
 * buffers dot[3n], color[3n] (the render-side consumer of the path).  The library also exports the
 * reference's own symbol `extern "C" void generate_dots(float3*, float3*, const std::shared_ptr<
 * SPHParticles>)` (vbo.cu:46-51; C++ callers declare it as main.cpp:268 does).                  */
int  sphx_generate_dots(const sphx_system *sys, float *device_dot, float *device_color);

/* Boundary particles for static obstacles (the step upstream of the path).  The reference
 * samples one shape, the shell of the domain box (main.cpp:89-116); sphx_create turns ANY boundary set into
 * masses with the reference's formula (computeBoundaryMass_CUDA, SPHSystem.cu:79-112).  These host-side,
 * deterministic samplers produce particle layers for other shapes, to be appended to the shell before
 * sphx_create.  Two-call protocol: out_xyz = NULL returns the count only.                          */
int  sphx_sample_box(const float lo[3], const float hi[3], float spacing, float *out_xyz, int capacity, int *count);
int  sphx_sample_sphere(const float center[3], float radius, float spacing, float *out_xyz, int capacity, int *count);
int  sphx_sample_triangles(const float *tri_xyz, int n_triangles, float spacing, float *out_xyz, int capacity, int *count);

/* State snapshots (checkpoint / resume and fixture I/O; the reference has none).
 * save: positions, velocities, ids, density, pressure in the CURRENT array order, the boundary set with
 * its masses, and the solver's persistent array (DFSPH warm stiffness / PBD last positions), in one
 * little-endian file.  load: creates a system that continues the saved run bit-identically: the
 * arrays are restored in the saved order and NOT re-sorted, so the next step's stable cell sort sees
 * exactly what the uninterrupted run would have seen.                                            */
int  sphx_snapshot_save(const sphx_system *sys, const char *path);
int  sphx_snapshot_load(const char *path, sphx_system **out);

#ifdef __cplusplus
}
#endif
#endif /* SPHX_C_H */
