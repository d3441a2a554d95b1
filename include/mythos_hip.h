/*
 * mythos_hip.h  --  C ABI of libmythos_hip.so: the MI355X (gfx950) force-evaluation and
 * Langevin-integrator core for oxDNA1/2 (and MARTINI 2/3) energy functions.
 *
 * The reference (mythos-bio/mythos) has no FFI: its plugin surface is the Python protocol
 *   EnergyFunction.__call__/map/with_params        mythos/energy/base.py:24-93, 215-434
 *   JaxMDSimulator.run -> scan(step_fn)            mythos/simulators/jax_md/jaxmd.py:45-103
 *   jax.value_and_grad over the energy             mythos/optimization/objective.py:198-235
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C, no torch / HIP types: streams are passed as void* (a hipStream_t, NULL = default).
 *   - "dev" pointers are device memory owned by the caller (PyTorch tensors) and only borrowed;
 *     "host" pointers are read during the call and may be freed afterwards.
 *   - dtype: 0 = float32 state/arithmetic, 1 = float64.  Energies and dU/dparams are always
 *     float64.
 *   - every function returning int returns 0 on success or a negative mythos_status; the text
 *     is available from mythos_last_error() (thread-local).
 *   - handles are independent: one handle per thread / stream.  The one piece of process-wide state is the table of
 *     test switches behind mythos_debug_set (tests only; all zero by default, never read per launch).
 *   - nucleotides are in oxDNA-classic 3'->5' memory order; quaternions are [w, x, y, z].
 */
#ifndef MYTHOS_HIP_H
#define MYTHOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mythos_system mythos_system_t;   /* one oxDNA system: topology + parameters + neighbours */
typedef struct mythos_sim mythos_sim_t;         /* Langevin integrator state bound to a system */
typedef struct mythos_martini mythos_martini_t; /* one MARTINI system */
typedef struct mythos_martini_sim mythos_martini_sim_t; /* Langevin integrator state bound to a MARTINI system */
typedef struct mythos_obs mythos_obs_t;         /* a set of per-frame structural observables (index lists on the device) */
typedef void* mythos_stream_t;                  /* hipStream_t */

enum mythos_status {
  MYTHOS_OK = 0,
  MYTHOS_ERR_INVALID_ARGUMENT = -1,
  MYTHOS_ERR_HIP = -2,
  MYTHOS_ERR_NO_DEVICE = -3,
  MYTHOS_ERR_NOT_READY = -4,   /* parameters or neighbours not set */
  MYTHOS_ERR_OVERFLOW = -5,    /* a static neighbour list too short, or more out-of-turn rebuilds in one run than a
                                  sane skin / rebuild interval produces (see mythos_langevin_last_recoveries) */
  MYTHOS_ERR_NUMERIC = -6      /* NaN / Inf detected */
};

enum mythos_dtype { MYTHOS_F32 = 0, MYTHOS_F64 = 1 };

/* number of energy terms reported per frame (dna1 leaves the Debye slot 0):
 * fene, bonded_excluded_volume, stacking, unbonded_excluded_volume, hydrogen_bonding,
 * cross_stacking, coaxial_stacking, debye  -- the order of dna2.default_energy_fns()
 * (mythos/energy/dna2/__init__.py:74-85). */
#define MYTHOS_OXDNA_N_TERMS 8

const char* mythos_version(void);
/* last error text of the calling thread ("" if none) */
const char* mythos_last_error(void);
/* number of visible HIP devices (0 if none); never initialises a context beyond the count */
int mythos_device_count(void);

/* ---- flat parameter vector ------------------------------------------------------------------
 * Replaces the per-term frozen dataclasses (*Configuration, mythos/energy/configuration.py:16-123)
 * at the kernel boundary: independent AND dependent constants, one double each, in the order
 * reported here.  The host-side shim derives the dependent ones (init_params) and applies the
 * chain rule to dU_dparams. */
int mythos_oxdna_param_count(void);
const char* mythos_oxdna_param_name(int index); /* NULL if out of range */

/* ---- system ---------------------------------------------------------------------------------
 * Replaces BaseEnergyFunction.__post_init__ topology capture (mythos/energy/base.py:133-140).
 *   model     1 = oxDNA1, 2 = oxDNA2, 3 = oxRNA2 (mythos/energy/rna2/: backbone site on a1 and a3, stacking between the
 *             3' / 5' sites with theta9 / theta10, cross-stacking without theta4, oxDNA1-form coaxial term, Debye-Hueckel)
 *             4 = oxNA, hybrid DNA / RNA systems (mythos/energy/na1/): every pair takes the parameter vector and the
 *             functional forms of its kind - DNA-DNA: oxDNA2, RNA-RNA: oxRNA2, DNA-RNA: the hybrid numbers in the oxDNA1
 *             forms - and every nucleotide the sites of its own type.  Such a system needs
 *             mythos_oxdna_set_nucleotide_types, takes 3 x mythos_oxdna_param_count() parameters (the oxDNA2, oxRNA2 and
 *             hybrid vectors one after the other; dU_dparams rows have the same layout).  No structural observables for it;
 *             a probabilistic sequence for hydrogen bonding only (mythos_oxdna_set_pseq terms = 2), with dU/d(distribution)
 *             through mythos_oxdna_energy_dpseq like the other models.  The Langevin integrator has an oxNA instantiation of its fused step kernel
 *             (two thirds of the oxDNA2 rate: a parameter set is chosen per row entry).  A probabilistic sequence and its gradient reach
 *             an oxNA system through the hydrogen-bonding term (mythos_oxdna_set_pseq, mythos_oxdna_energy_dpseq).
 *   seq       host int32[n]  bases A,C,G,T -> 0..3 (mythos/utils/constants.py:5-11)
 *   is_end    host uint8[n]  1 for strand-terminal nucleotides (Debye half charges), may be NULL
 *   bonded    host int32[n_bonded][2] rows (nn_i, nn_j) as mythos/input/topology.py:166-183
 *   box       host double[3] periodic box (jax_md.space.periodic), or NULL for free space
 */
mythos_system_t* mythos_oxdna_create(int model, int n, const int32_t* seq, const uint8_t* is_end, int n_bonded,
                                     const int32_t* bonded, const double* box, int dtype, int device);
void mythos_oxdna_destroy(mythos_system_t* sys);

/* host double[n_params] in mythos_oxdna_param_name() order (oxNA: three such vectors); replaces with_params() at the boundary */
int mythos_oxdna_set_params(mythos_system_t* sys, const double* flat_params, int n_params);

/* oxNA (model 4) only: which nucleotides are RNA - the `nt_type` every na1 *Configuration carries
 * (mythos/energy/na1/fene.py:22, mythos/input/topology.py NucleotideType; is_rna_pair / is_dna_rna_pair in
 * mythos/energy/na1/utils.py:9-16).  host uint8[n], 1 = RNA, 0 = DNA. */
int mythos_oxdna_set_nucleotide_types(mythos_system_t* sys, const uint8_t* is_rna);

/* Probabilistic sequence: replaces `pseq` / `pseq_constraints` of StackingConfiguration and
 * HydrogenBondingConfiguration (mythos/energy/dna1/stacking.py:54-55, 261-287, dna1/hydrogen_bonding.py:94-95, 308-333)
 * and compute_seq_dep_weight (mythos/energy/utils.py:45-132).  The sequence-dependent weight of a pair becomes its
 * expectation; the energy entry point (energies, forces, dU/dparams incl. the 4x4 weight tables) honours it.
 *   marginals  host double[n][4]   probability of A, C, G, T per nucleotide (for a base-paired nucleotide: the marginal
 *              of its pair's type distribution)
 *   unit       host int32[n]       2 * base_pair + position_in_pair for constrained base pairs, -1 for unpaired
 *   bp_probs   host double[n_bp][4] probability of the types AT, TA, GC, CG (mythos/utils/constants.py:13) per base pair
 *   terms      bit 0: stacking, bit 1: hydrogen bonding use the expectation; 0 switches back to the discrete sequence
 * The Langevin integrator steps such a system too (since round 3): the stepping kernel has instantiations whose two
 * weight look-ups are the expectations (md_step_kernel<..., PSEQ>), as the reference's configurations carry pseq into
 * any energy function, the one a simulator steps with included (dna1/stacking.py:284-285, hydrogen_bonding.py:330-331). */
int mythos_oxdna_set_pseq(mythos_system_t* sys, const double* marginals, const int32_t* unit, int n_bp,
                          const double* bp_probs, int terms);

/* dU/d(sequence distribution): what jax.grad of the reference's energy function returns for the `pseq` leaves
 * (compute_seq_dep_weight is differentiable in them, mythos/energy/utils.py:45-132).  The energy call with parameter
 * partials, plus, per frame,
 *   dU_dmarginals  device double[n_frames][n][4]        dU/d(marginal base probabilities) through the pairs whose two
 *                                                        nucleotides are independent (different units)
 *   dU_dbp         device double[n_frames][max(n_bp,1)][4]  dU/d(type probabilities) through the pairs that ARE a
 *                                                        constrained base pair
 * in the layout mythos_oxdna_set_pseq took; the host carries them to the reference's (unpaired, base-pair) arrays - a
 * paired nucleotide's marginal is a sum of its pair's type probabilities.  Accumulated with fp64 atomics: equal to
 * rounding between runs, not bit for bit.  Needs mythos_oxdna_set_pseq with terms != 0. */
int mythos_oxdna_energy_dpseq(mythos_system_t* sys, const void* center, const void* quat, int n_frames, double* e_terms,
                              void* dU_dcenter, void* dU_dquat, double* dU_dparams, double* dU_dmarginals, double* dU_dbp,
                              mythos_stream_t stream);

/* Unbonded pair list, reference semantics (NoNeighborList.idx, simulators/jax_md/utils.py:48-67):
 * host int32[n_pairs][2], rows (op_i, op_j) in that role order.  Converted to per-nucleotide rows. */
int mythos_oxdna_set_neighbors(mythos_system_t* sys, const int32_t* pairs, int n_pairs);

/* GPU Verlet list with bonded exclusions from positions (replaces mythos/utils/neighbors.py:12-59):
 * center dev real[n][3]; pairs with |c_i - c_j| < r_cut + skin are kept. */
int mythos_oxdna_build_neighbors(mythos_system_t* sys, const void* center, double r_cut, double skin,
                                 mythos_stream_t stream);
/* max and mean row length of the current list (directed neighbours per nucleotide) */
int mythos_oxdna_neighbor_stats(mythos_system_t* sys, int* max_row, double* mean_row);

/* Energy of n_frames configurations; replaces ComposedEnergyFunction.compute_terms / map
 * (mythos/energy/base.py:312-319, 90-93) and the jax.grad of them.
 *   center      dev real[n_frames][n][3]
 *   quat        dev real[n_frames][n][4]
 *   e_terms     dev double[n_frames][8]                      (required)
 *   dU_dcenter  dev real[n_frames][n][3] or NULL
 *   dU_dquat    dev real[n_frames][n][4] or NULL             (gradient w.r.t. the un-normalised quaternion,
 *                                                            as jax.grad of mythos/energy/utils.py:18-36 gives)
 *   dU_dparams  dev double[n_frames][n_params] or NULL       (flat vector; dependent entries treated as free)
 */
int mythos_oxdna_energy(mythos_system_t* sys, const void* center, const void* quat, int n_frames, double* e_terms,
                        void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_stream_t stream);

/* ---- per-frame structural observables ------------------------------------------------------------
 * Replaces mythos/observables/propeller.py:19-71, pitch.py:33-102, rise.py:21-80 and the per-state part of
 * persistence_length.py:47-91, 168-185 (base.py:24-66 for the local helical axis and the quartets).
 *   geometry    host double[3]: com_to_hb, backbone offset along a1, backbone offset along a2 (0 for oxDNA1; for
 *               oxRNA2, model 3, the offset along a3)
 *   box         host double[3] periodic box of the displacement function, or NULL (free space)
 *   base_pairs  host int32[n_bp][2]       hydrogen-bonded pairs of the propeller twist
 *   quartets    host int32[n_q][2][2]     adjacent base pairs ((a1, b1), (a2, b2)) of rise / pitch / persistence length
 *   skip_ends   drop two quartets at either end in the persistence-length partials (persistence_length.py:84-87)
 * Output row per frame, mythos_observables_width() = 4 + n_corr doubles (n_corr = n_q - 4 with skip_ends, else n_q):
 *   [0] propeller twist (deg)  [1] rise (Angstrom)  [2] pitch angle (rad)  [3] mean base-pair spacing <l0>
 *   [4 + d] autocorrelation C(d) of the local helical axes.   Entries whose list is empty are 0.
 * mythos_observables_eval is the stand-alone launch; mythos_oxdna_energy_obs returns the same rows with the energies - the
 * same kernel, queued behind the energy launch of that call while the frames are still in L2 (energies, dU/dparams and
 * observables from one call: the DiffTRe data path). */
mythos_obs_t* mythos_observables_create(int model, int n, const double* geometry, const double* box, int n_bp,
                                        const int32_t* base_pairs, int n_quartets, const int32_t* quartets, int skip_ends,
                                        int dtype, int device);
void mythos_observables_destroy(mythos_obs_t* obs);
int mythos_observables_width(const mythos_obs_t* obs);
/* center dev real[n_frames][n][3], quat dev real[n_frames][n][4], out dev double[n_frames][width] */
int mythos_observables_eval(mythos_obs_t* obs, const void* center, const void* quat, int n_frames, double* out,
                            mythos_stream_t stream);
/* mythos_oxdna_energy with observables in the same launch: obs may be NULL (then identical to mythos_oxdna_energy);
 * obs_out dev double[n_frames][width] */
int mythos_oxdna_energy_obs(mythos_system_t* sys, const void* center, const void* quat, int n_frames, double* e_terms,
                            void* dU_dcenter, void* dU_dquat, double* dU_dparams, mythos_obs_t* obs, double* obs_out,
                            mythos_stream_t stream);

/* ---- Langevin integrator ----------------------------------------------------------------------
 * Replaces jax_md.simulate.nvt_langevin on RigidBody states as driven by
 * mythos/simulators/jax_md/jaxmd.py:73-94: BAOAB, rigid-body rotation with body-frame angular
 * momentum (equivalent to the quaternion-momentum NO_SQUISH form), counter-based RNG.
 *   gamma_t/gamma_r   friction of the translational / rotational thermostat
 *   inertia           host double[3] principal moments
 */
mythos_sim_t* mythos_langevin_create(mythos_system_t* sys, double dt, double kT, double gamma_t, double gamma_r,
                                     double mass, const double* inertia, uint64_t seed);
void mythos_langevin_destroy(mythos_sim_t* sim);

/* neighbour-list policy of the MD loop: rebuild every `every` steps with cut-off r_cut + skin
 * (every <= 0: keep the list given by set_neighbors for the whole run) */
int mythos_langevin_set_neighbor_policy(mythos_sim_t* sim, double r_cut, double skin, int every);

/* draw Maxwell-Boltzmann momenta at kT (init_fn of nvt_langevin) */
int mythos_langevin_init_momenta(mythos_sim_t* sim, void* p_lin, void* p_ang, mythos_stream_t stream);

/* Run n_steps; state arrays are updated in place.
 *   center dev real[n][3], quat dev real[n][4], p_lin dev real[n][3], p_ang dev real[n][3] (body frame)
 *   save_every > 0: traj_center dev real[n_steps/save_every][n][3], traj_quat [..][n][4] (aligned to 4 elements)
 *   receive the state after steps save_every, 2*save_every, ...; e_trace dev double[n_steps/save_every][10] receives
 *   the 8 term energies + translational + rotational kinetic energy (any of the three may be NULL).
 *   e_trace == NULL is the reference's own run (mythos/simulators/jax_md/jaxmd.py:84-99 stores state.position of every
 *   step and nothing else): the step launch that produces a saved state writes it to its row too - 7 more words per
 *   nucleotide and saved step, no other cost.  e_trace != NULL selects the energy-trace instantiation of the step
 *   kernel for the saved steps plus one small reduction launch per row.
 * A site that outruns the Verlet skin before the scheduled rebuild, or rows / cell buckets that outgrow their
 * allocation, halt the queued launches; the run rebuilds at the last valid state and resumes (not an error; counted
 * by mythos_langevin_last_recoveries).  MYTHOS_ERR_OVERFLOW only after 64 such rebuilds in one run, or when one
 * nucleotide has more than 32 neighbours inside the range of an angular term (up to 16 the fast instantiation runs; a
 * launch that finds more aborts, nothing it wrote counts, and the step is repeated with 32 result rows per nucleotide).
 * After an error the arrays hold the positions of the last step that counted and momenta short of that step's closing
 * half kick; mythos_langevin_get_step tells how many steps that was. */
int mythos_langevin_run(mythos_sim_t* sim, void* center, void* quat, void* p_lin, void* p_ang, int n_steps,
                        int save_every, void* traj_center, void* traj_quat, double* e_trace,
                        mythos_stream_t stream);
/* Resident form of the same loop: the state stays in the integrator's own layout on the device between calls, as
 * the carry of the reference's jax.lax.scan stays on its device (mythos/simulators/jax_md/jaxmd.py:87-94).
 *   load     caller's arrays -> resident state (asynchronous); the neighbour list is rebuilt at the next advance
 *   advance  n_steps on the resident state; the list and its rebuild schedule carry over from the previous advance;
 *            outputs as mythos_langevin_run.  One stream synchronisation per call (the halt-and-resume protocol
 *            needs the host to see the control words before the steps count as taken), nothing else between calls
 *   store    resident state -> caller's arrays (asynchronous on the stream); the state stays resident
 * mythos_langevin_run(...) == load; advance; store.  advance after advance continues bit for bit like one advance
 * of the summed length.  An advance that ends in MYTHOS_ERR_NUMERIC drops the resident state.
 * One force evaluation per step: advance(n) is n launches of the step kernel and leaves the resident momenta short of
 * the closing half kick of step n, which needs the forces at x_n - the evaluation the next advance starts with anyway
 * (as the reference's carry holds the force of the last step for the next one, jax_md simulate.nvt_langevin).  store
 * supplies it with one more launch when the frame is still open, so what store hands back is always (x_n, p_n); an
 * advance whose last step saves an ENERGY row (e_trace != NULL) closes while it evaluates that row.  An open frame is
 * closed with the parameters / sequence distribution in force WHEN IT IS CLOSED: a caller that replaces them between
 * two advances (mythos_oxdna_set_params, mythos_oxdna_set_pseq) and wants step n finished under the old ones calls store
 * first.  The mythos_langevin_last_* figures
 * describe the last call that launched step kernels, store's closing launch included. */
int mythos_langevin_load(mythos_sim_t* sim, const void* center, const void* quat, const void* p_lin, const void* p_ang,
                         mythos_stream_t stream);
int mythos_langevin_advance(mythos_sim_t* sim, int n_steps, int save_every, void* traj_center, void* traj_quat,
                            double* e_trace, mythos_stream_t stream);
int mythos_langevin_store(mythos_sim_t* sim, void* center, void* quat, void* p_lin, void* p_ang, mythos_stream_t stream);

/* absolute step counter (RNG stream position); settable for checkpoint/resume */
int64_t mythos_langevin_get_step(const mythos_sim_t* sim);
int mythos_langevin_set_step(mythos_sim_t* sim, int64_t step);
/* the key of the noise (Philox key; init_momenta draws from it too): what a new `key` argument of the reference's
 * run(opt_params, init_state, n_steps, key) is (mythos/simulators/jax_md/jaxmd.py:60-68), without a new integrator */
int mythos_langevin_set_seed(mythos_sim_t* sim, uint64_t seed);

/* Integrator options.  MYTHOS_LANGEVIN_UNFUSED (oxNA systems only, value 0 / 1): step through the two-launch path -
 * the energy kernel's forces launch + a one-thread-per-nucleotide integrator - instead of the fused step kernel.  A
 * second implementation of the same map (same Philox stream), kept as the cross-check of the fused oxNA instantiation;
 * it has no halt-and-resume (a skin violation is an error).  Takes effect at the next mythos_langevin_load / _run and
 * holds while that state is resident. */
enum mythos_langevin_option { MYTHOS_LANGEVIN_UNFUSED = 0 };
int mythos_langevin_set_option(mythos_sim_t* sim, int option, int64_t value);

/* timing hook for bench.py, HIP events on the launch stream of the last run / advance (all zero unless
 * mythos_langevin_set_timing asked for samples: an untimed run records no events at all):
 *   kernel_ms            mean duration of the step kernel over up to 16 launches spread evenly over the
 *                        run, each timed by its own event pair attached to the dispatch
 *                        (hipExtLaunchKernelGGL start/stop events = the dispatch's begin/end stamps)
 *   loop_ms_per_launch   (last event - first event) / launches: includes neighbour rebuilds and
 *                        inter-kernel gaps */
/* samples: how many dispatches of a run get their own event pair (0 = none, the default; at most 16).  A bracketed
 * dispatch costs ~8 us of queue time, so timing is something the caller switches on (bench.py does). */
int mythos_langevin_set_timing(mythos_sim_t* sim, int samples);
int mythos_langevin_last_kernel_ms(const mythos_sim_t* sim, double* kernel_ms, double* loop_ms_per_launch,
                                   int* launches, int* samples);

/* How often the last run halted and resumed: a site left its skin before the scheduled rebuild, or a rebuild
 * overflowed its rows / cell buckets.  The launches behind such an event do nothing; the run rebuilds at the last
 * valid state (growing what overflowed) and continues there, so neither is an error - the reference's neighbour
 * list signals `did_buffer_overflow` and leaves the reallocation to the caller (jax_md partition, used at
 * mythos/simulators/jax_md/utils.py:70-126).  More than 64 in one run fails with MYTHOS_ERR_OVERFLOW. */
int mythos_langevin_last_recoveries(const mythos_sim_t* sim, int* recoveries);
/* scheduled list rebuilds (every rebuild_every steps) that fell inside the last run / advance; the build that creates a
 * list and the out-of-turn ones above are not counted.  bench.py reports it next to the timed region. */
int mythos_langevin_last_rebuilds(const mythos_sim_t* sim, int* scheduled);

/* ---- oxDNA text trajectories (host only) -----------------------------------------------------
 * Replaces the Python parse of mythos/input/trajectory.py:192-320 (frames of `t = / b = / E =` header lines and
 * n rows of 15 numbers: com, a1, a3, v, L).  Call once with frames == NULL to count (n_frames out), then with
 * host buffers times[F], box[F][3], energies[F][3], frames[F][n][15]; at most max_frames are stored.  Rows stay in
 * file order (the 5'->3' reversal of new-format files is the caller's, trajectory.py:309-313). */
int mythos_oxdna_read_trajectory(const char* path, int n, int max_frames, double* times, double* box, double* energies,
                                 double* frames, int* n_frames);

/* Writer for the same format (mythos/input/trajectory.py:322-331, mythos/simulators/io.py:146-170): host buffers
 * times[F], box[F][3], energies[F][3], frames[F][n][15]; every number as the shortest text that parses back to the
 * same double (what the reference's str(float) prints); append != 0 adds to an existing file. */
int mythos_oxdna_write_trajectory(const char* path, int n, int n_frames, const double* times, const double* box,
                                  const double* energies, const double* frames, int append);

/* ---- test and diagnostic switches ---------------------------------------------------------------
 * No counterpart in the reference.  The GPU tests use them to reach code paths that physical inputs reach only by
 * chance (a cell bucket that overflows, a row walk in several segments, a list rebuild that overflows in front of the
 * first launch of a segment).  Process-wide, read when a list is built / a call starts (never per launch); a value of 0
 * restores the default.  Nothing in the library reads the environment. */
enum mythos_debug_key {
  MYTHOS_DEBUG_CELL_BUCKET_CAP = 0, /* places per cell bucket, fixed (no growth): exercises the spill list */
  MYTHOS_DEBUG_ENERGY_LIST_CAP = 1, /* entries per segment of the energy kernel's row walk (8 ... 192) */
  MYTHOS_DEBUG_MD_SEGMENT = 2,      /* step launches queued between two looks at the halt word (default 8192) */
  MYTHOS_DEBUG_MD_OVERFLOW_AT = 3,  /* k + 1: the scheduled list rebuild in front of step launch k reports a row
                                       overflow although its rows fit (one shot: cleared when it fires) */
  MYTHOS_DEBUG_MD_DENSE = 4,        /* 1: every fp32 oxDNA1/2 stepping launch takes the DENSE instantiation (normally
                                       grids of more than four workgroups per CU), 2: none does */
  MYTHOS_DEBUG_MD_ITEMS_BIG = 5,    /* 1: oxDNA step launches start on the wide work lists (ITEMS = 32) instead of reaching them
                                       through an aborted launch */
  MYTHOS_DEBUG_MD_LANES = 6,        /* 8 | 16: lanes per nucleotide of the oxDNA step launches, fixed when a state is loaded
                                       (normally 16 where n / 16 workgroups are at most two per CU, 8 otherwise) */
  MYTHOS_DEBUG_KEYS = 7
};
int mythos_debug_set(int key, int64_t value);
int64_t mythos_debug_get(int key);

/* ---- MARTINI 2/3 ------------------------------------------------------------------------------
 * Replaces mythos/energy/martini/m2/{lj,bond,angle}.py and m3/angle.py.
 *   types        host int32[n]            bead type index
 *   sigma, eps   host double[n_types][n_types]
 *   bonds        host int32[n_bonds][2], bond_k / bond_r0 host double[n_bonds]
 *   angles       host int32[n_angles][3], angle_k / angle_t0 host double[n_angles] (t0 in radians)
 *   angle_kind   0 = G96 cosine (MARTINI 2), 1 = harmonic (MARTINI 3)
 *   r_cut        LJ cut-off (1.1 nm in the reference), potential shifted to 0 at r_cut
 * Energy terms per frame: [lj, bond, angle].  box dev real[n_frames][3] (per-frame periodic box).
 */
#define MYTHOS_MARTINI_N_TERMS 3
mythos_martini_t* mythos_martini_create(int n, const int32_t* types, int n_types, const double* sigma,
                                        const double* eps, int n_bonds, const int32_t* bonds, const double* bond_k,
                                        const double* bond_r0, int n_angles, const int32_t* angles,
                                        const double* angle_k, const double* angle_t0, int angle_kind, double r_cut,
                                        int dtype, int device);
void mythos_martini_destroy(mythos_martini_t* m);
int mythos_martini_energy(mythos_martini_t* m, const void* pos, const void* box, int n_frames, double* e_terms,
                          void* dU_dpos, mythos_stream_t stream);

/* Parameter gradients per frame (the reference: jax.grad of the same energy functions with respect to the
 * configuration values, mythos/energy/martini/m2/lj.py:137-157, bond.py:60-71, angle.py:113-129).
 * All outputs are dev double and optional (NULL = skip), except that d_sigma and d_eps come as a pair:
 *   d_sigma, d_eps   [n_frames][n_types][n_types]  dU_lj/dsigma[a][b], dU_lj/deps[a][b] for the ORDERED type
 *                    pair (owner, partner), each unordered bead pair contributing half to either order: the
 *                    derivative with respect to a symmetric entry is out[a][b] + out[b][a]
 *   d_bond_k/_r0     [n_frames][n_bonds]     d_angle_k/_t0  [n_frames][n_angles]   (one value per bond / angle;
 *                    the host sums the bonds that share a named parameter) */
int mythos_martini_param_grads(mythos_martini_t* m, const void* pos, const void* box, int n_frames, double* d_sigma,
                               double* d_eps, double* d_bond_k, double* d_bond_r0, double* d_angle_k,
                               double* d_angle_t0, mythos_stream_t stream);

/* ---- MARTINI Langevin MD ----------------------------------------------------------------------
 * BASELINE configs[2].  The reference runs MARTINI dynamics in GROMACS as an external process
 * (mythos/simulators/gromacs/) and has no integrator of its own for it; this is the device-resident
 * integrator for the same force field (LJ over a Verlet list rebuilt on the device, bonds, angles) with the
 * BAOAB Langevin splitting of mythos_langevin_run for point particles.  Units: nm, ps, amu, kJ/mol.
 *   gamma    friction rate 1/ps (GROMACS sd integrator: 1 / tau_t);  mass host double[n] or NULL (72 amu)
 *   pos, vel dev real[n][3], updated in place;  box host double[3] (orthorhombic, fixed during the run)
 *   traj_pos dev real[n_steps/save_every][n][3] or NULL;  e_trace dev double[.][4] = lj, bond, angle, kinetic
 * Errors and the halt-and-resume protocol as mythos_langevin_run (NUMERIC: NaN). */
mythos_martini_sim_t* mythos_martini_langevin_create(mythos_martini_t* sys, double dt, double kT, double gamma,
                                                     const double* mass, uint64_t seed);
void mythos_martini_langevin_destroy(mythos_martini_sim_t* sim);
int mythos_martini_langevin_set_neighbor_policy(mythos_martini_sim_t* sim, double skin, int rebuild_every);
/* Pruned rows (round 4; no counterpart in the reference - GROMACS, which runs its MARTINI dynamics, prunes its pair list
 * between searches in the same way): every `every` steps the step launch writes, as a by-product of the distances it
 * computes, the entries of each Verlet row inside r_cut + margin to a second set of rows, and the launches in between walk
 * those.  Safe while no bead moves more than margin / 2 between two prunings: checked per step (a step longer than
 * margin / (2 (every - 1)) halts and rebuilds like a bead that leaves its skin).  Off by default; bench.py runs the
 * bilayer with margin 0.2 nm, every 4.  margin <= 0, margin >= skin or every < 2 switches the pruned rows off.  The forces are those of the Verlet rows up to the
 * order of the sums. */
int mythos_martini_langevin_set_inner_list(mythos_martini_sim_t* sim, double margin, int every);
int mythos_martini_langevin_init_velocities(mythos_martini_sim_t* sim, void* vel, mythos_stream_t stream);
int mythos_martini_langevin_run(mythos_martini_sim_t* sim, void* pos, void* vel, const double* box, int n_steps,
                                int save_every, void* traj_pos, double* e_trace, mythos_stream_t stream);
/* Resident form, as mythos_langevin_load / advance / store: the state stays in the integrator's layout on the device
 * between calls, the list and its rebuild schedule carry over, advance(n) is n launches of the step kernel and leaves the
 * frame open (velocities short of the closing half kick of step n; the next advance or store supplies it).
 * mythos_martini_langevin_run == load; advance; store.  advance(a); advance(b) == advance(a + b), bit for bit.
 * e_trace == NULL with traj_pos != NULL: positions-only rows, written by the step launch that produces the saved state. */
int mythos_martini_langevin_load(mythos_martini_sim_t* sim, const void* pos, const void* vel, const double* box,
                                 mythos_stream_t stream);
int mythos_martini_langevin_advance(mythos_martini_sim_t* sim, int n_steps, int save_every, void* traj_pos, double* e_trace,
                                    mythos_stream_t stream);
int mythos_martini_langevin_store(mythos_martini_sim_t* sim, void* pos, void* vel, mythos_stream_t stream);
int64_t mythos_martini_langevin_get_step(const mythos_martini_sim_t* sim);
/* scheduled list rebuilds inside the last advance (as mythos_langevin_last_rebuilds) */
int mythos_martini_langevin_last_rebuilds(const mythos_martini_sim_t* sim, int* scheduled);
int mythos_martini_langevin_last_kernel_ms(const mythos_martini_sim_t* sim, double* kernel_ms,
                                           double* loop_ms_per_launch, int* launches, int* samples);
int mythos_martini_langevin_set_timing(mythos_martini_sim_t* sim, int samples); /* as mythos_langevin_set_timing */
/* out-of-turn rebuilds of the last run (same protocol as mythos_langevin_last_recoveries) */
int mythos_martini_langevin_last_recoveries(const mythos_martini_sim_t* sim, int* recoveries);
int mythos_martini_langevin_neighbor_stats(const mythos_martini_sim_t* sim, int* max_row, double* mean_row);
/* Diagnostics / tests: the neighbour rows as the device holds them, which = 0 the Verlet rows, 1 the pruned rows
 * (mythos_martini_langevin_set_inner_list).  rows host int32[n][*stride], row_len host int32[n]; with both NULL only
 * *stride is set.  Synchronises the device. */
int mythos_martini_langevin_get_rows(const mythos_martini_sim_t* sim, int which, int32_t* rows, int32_t* row_len, int* stride);

#ifdef __cplusplus
}
#endif
#endif /* MYTHOS_HIP_H */
