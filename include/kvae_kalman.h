/*
 * kvae_kalman.h -- C ABI of the MI355X (gfx950) Kalman filter / RTS smoother /
 * Kalman-ELBO kernels.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream
 * (passed as void*); nothing here allocates, synchronises with the host or
 * touches a torch type.  All tensors use the reference's layout: sequence
 * first, row-major contiguous, e.g. means (T,B,n), covariances (T,B,n,n).
 * "real" below means float when dims.dtype == KVAE_F32 and double when
 * KVAE_F64; every pointer of one call has that element type.
 *
 * Return value: KVAE_OK or a negative KVAE_E* code (kvae_error_string()).
 * Numerical failures (a non-positive Cholesky pivot) do not abort a launch;
 * they OR a bit into the int32 word `status` points to (may be NULL).
 *
 * The reference has no FFI layer; the interface each entry point replaces is
 * a Python method of kvae/state_space_model.py or a block of
 * kvae/kalman_vae.py and is cited on the declaration.
 */
#ifndef KVAE_KALMAN_H
#define KVAE_KALMAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVAE_ABI_VERSION 1

enum { KVAE_F32 = 0, KVAE_F64 = 1 };

enum {
    KVAE_OK = 0,
    KVAE_E_BADARG = -1,      /* null required pointer, T < 1, B < 1 ...          */
    KVAE_E_UNSUPPORTED = -2, /* (n,p,K,dtype) has no compiled kernel             */
    KVAE_E_WORKSPACE = -3,   /* workspace smaller than kvae_workspace_bytes()    */
    KVAE_E_LAUNCH = -4,      /* hipGetLastError() after the launch was not ok    */
    KVAE_E_TOOLARGE = -5     /* beyond the fixed scratch of a *_noalloc entry point */
};

/* bits of the device status word */
enum {
    KVAE_STATUS_NONPD_INNOVATION = 1, /* S = C P C^T + R had a pivot <= 0        */
    KVAE_STATUS_NONPD_PREDICTED = 2,  /* predicted covariance (smoother inverse)  */
    KVAE_STATUS_NONPD_SMOOTHED = 4,   /* smoothed covariance (sampling / ELBO)    */
    KVAE_STATUS_NONPD_NOISE = 8       /* Q, R or the initial covariance           */
};

typedef struct {
    int32_t T;          /* steps in this call (a segment of a longer sequence is allowed) */
    int32_t B;          /* sequences                                             */
    int32_t n;          /* state dimension  (z_dim)                              */
    int32_t p;          /* observation dimension (a_dim)                         */
    int32_t K;          /* number of basis matrices                              */
    int32_t dtype;      /* KVAE_F32 / KVAE_F64                                   */
    int32_t symmetrize; /* symmetrize_covariance of the reference                */
    int32_t burn_in;    /* steps t < burn_in pass no gradient (backward only)    */
} kvae_dims;

/* ------------------------------------------------------------------------ */
/* Filter: StateSpaceModel.kalman_filter, kvae/state_space_model.py:172-373
 * (the recursion of lines 284-352; the imputation sample of lines 232-277 and
 * the weight network of lines 279-282 stay on the PyTorch side).
 *
 * For t = 0..T-1, per sequence b, with A_t = sum_k alpha[t,b,k] A_K[k] and
 * C_t likewise: measurement update with C_t, mask blend, symmetrise, time
 * update with A_{t+1}, symmetrise.
 * The recursion starts from (m_in, P_in): shape (n)/(n,n) shared by all
 * sequences when carry_per_sequence == 0 (initial_state_mean/covariance), or
 * (B,n)/(B,n,n) when 1 (continuing a sequence: pass the previous call's
 * mean_pred[T-1], cov_pred[T-1]).
 */
typedef struct {
    const void *a;        /* (T,B,p)   observations                              */
    const void *alpha;    /* (T+1,B,K) mixing weights, row t+1 = weights after a_t */
    const void *mask;     /* (T,B) real in [0,1], or NULL = all observed         */
    const void *A_K;      /* (K,n,n)                                             */
    const void *C_K;      /* (K,p,n)                                             */
    const void *Q;        /* (n,n)                                               */
    const void *R;        /* (p,p)                                               */
    const void *m_in;     /* see above                                           */
    const void *P_in;
    int32_t carry_per_sequence;
    int32_t _pad;
    void *mean_filt;      /* (T,B,n)    z_{t|t}                                  */
    void *cov_filt;       /* (T,B,n,n)                                           */
    void *mean_pred;      /* (T,B,n)    z_{t+1|t}                                */
    void *cov_pred;       /* (T,B,n,n)                                           */
    void *mat_As;         /* (T+1,B,n,n) or NULL                                 */
    void *mat_Cs;         /* (T+1,B,p,n) or NULL                                 */
    int32_t *status;      /* or NULL                                             */
} kvae_filter_fwd;

int kvae_filter_forward(const kvae_dims *d, const kvae_filter_fwd *io, void *stream);

/* Adjoint of kvae_filter_forward (the reference differentiates lines 172-373
 * with the autograd tape; there is no hand-written counterpart).  Needs the
 * forward's inputs and its four mean/cov outputs.  Upstream gradients may be
 * NULL (= zero).  g_carry_* are the gradients w.r.t. mean_pred[T-1] /
 * cov_pred[T-1] arriving from a following segment.  Parameter gradients are
 * overwritten, not accumulated.
 */
typedef struct {
    kvae_filter_fwd fwd;          /* same pointers as the forward call           */
    const void *g_mean_filt, *g_cov_filt, *g_mean_pred, *g_cov_pred;
    const void *g_mat_As, *g_mat_Cs;   /* (T+1,...) */
    void *d_a;            /* (T,B,p)                                             */
    void *d_alpha;        /* (T+1,B,K)                                           */
    void *d_m_in;         /* (B,n)   per-sequence, also when the carry is shared */
    void *d_P_in;         /* (B,n,n)                                             */
    void *d_A_K;          /* (K,n,n)  summed over T and B                        */
    void *d_C_K;          /* (K,p,n)                                             */
    void *d_Q;            /* (n,n)                                               */
    void *d_R;            /* (p,p)                                               */
} kvae_filter_bwd;

int kvae_filter_backward(const kvae_dims *d, const kvae_filter_bwd *io,
                         void *workspace, size_t workspace_bytes, void *stream);

/* kvae_filter_backward for callers that cannot allocate device memory or pass a
 * workspace: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  It has no counterpart in the reference.  Same arguments and results as
 * kvae_filter_backward (bit for bit), but the per-wavefront partial sums of the parameter
 * gradients live in a fixed array of KVAE_NOALLOC_SCRATCH_BYTES inside the library's code
 * object (one copy per GPU, held until the process ends) instead of a workspace.  Calls no
 * hipMalloc / hipFree or any other allocating routine.  Returns KVAE_E_TOOLARGE when
 * kvae_workspace_bytes(d, KVAE_OP_FILTER_BWD) exceeds the array (at n=4, p=2, K=3, float:
 * up to 182 000 sequences; large-state shapes, whose workspace grows with T*B*n*n, only
 * for small batches).  All calls on one GPU share the array: two calls that overlap in
 * time on the same GPU (different streams) race; calls on one stream are safe.  The first
 * call on a device looks the array's address up (hipGetSymbolAddress); make it outside
 * graph capture.                                                                        */
#define KVAE_NOALLOC_SCRATCH_BYTES (4u << 20)
size_t kvae_noalloc_scratch_bytes(void);
int kvae_filter_backward_noalloc(const kvae_dims *d, const kvae_filter_bwd *io, void *stream);

/* ------------------------------------------------------------------------ */
/* Smoother: StateSpaceModel.kalman_smooth, kvae/state_space_model.py:375-484.
 * Reverse-time RTS recursion plus the per-step draw z_t = mean + chol(cov) eps
 * and re-emission a_t = C_t z_t + chol(R) eps'.  eps_state / eps_obs are
 * standard-normal draws indexed by time step, NULL selects the "mean" mode.
 * LR = chol(R) is only read when eps_obs != NULL.
 */
typedef struct {
    const void *mean_filt, *cov_filt, *mean_pred, *cov_pred; /* filter outputs  */
    const void *mat_As;   /* (T+1,B,n,n)                                         */
    const void *mat_Cs;   /* (T+1,B,p,n)                                         */
    const void *LR;       /* (p,p) lower Cholesky factor of R, or NULL           */
    const void *eps_state;/* (T,B,n) or NULL                                     */
    const void *eps_obs;  /* (T,B,p) or NULL                                     */
    void *mean_smooth;    /* (T,B,n)                                             */
    void *cov_smooth;     /* (T,B,n,n)                                           */
    void *zs;             /* (T,B,n)                                             */
    void *as_resampled;   /* (T,B,p)                                             */
    int32_t *status;
} kvae_smooth_fwd;

int kvae_smooth_forward(const kvae_dims *d, const kvae_smooth_fwd *io, void *stream);

typedef struct {
    kvae_smooth_fwd fwd;
    const void *g_mean_smooth, *g_cov_smooth, *g_zs, *g_as_resampled; /* or NULL */
    void *d_mean_filt, *d_cov_filt, *d_mean_pred, *d_cov_pred;        /* (T,B,..) */
    void *d_mat_As;       /* (T+1,B,n,n)                                         */
    void *d_mat_Cs;       /* (T+1,B,p,n)                                         */
    void *d_LR;           /* (p,p) summed over T and B, or NULL                  */
} kvae_smooth_bwd;

int kvae_smooth_backward(const kvae_dims *d, const kvae_smooth_bwd *io,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* Kalman ELBO terms: kvae/kalman_vae.py:165-168,191-242, before `aggregate`.
 * z_t = mean_smooth_t + chol(cov_smooth_t) eps_t;
 * obs[t,b]   = log N(a_t; C_t z_t, R)                 (C_0..C_{T-1})
 * trans[0,b] = log N(z_0; m0, P0); trans[t,b] = log N(z_t; A_t z_{t-1}, Q)
 * post[t,b]  = log N(z_t; mean_smooth_t, cov_smooth_t)
 */
typedef struct {
    const void *a;           /* (T,B,p)                                          */
    const void *mean_smooth; /* (T,B,n)                                          */
    const void *cov_smooth;  /* (T,B,n,n)                                        */
    const void *mat_As;      /* (T+1,B,n,n)                                      */
    const void *mat_Cs;      /* (T+1,B,p,n)                                      */
    const void *Q, *R;       /* (n,n), (p,p)                                     */
    const void *m0, *P0;     /* (n), (n,n)                                       */
    const void *eps;         /* (T,B,n)                                          */
    void *obs, *trans, *post;/* (T,B) each                                       */
    void *z;                 /* (T,B,n) the sample, or NULL                      */
    int32_t *status;
} kvae_elbo_fwd;

int kvae_elbo_forward(const kvae_dims *d, const kvae_elbo_fwd *io, void *stream);

typedef struct {
    kvae_elbo_fwd fwd;
    const void *g_obs, *g_trans, *g_post; /* (T,B) upstream gradients, or NULL   */
    void *d_a;            /* (T,B,p)                                             */
    void *d_mean_smooth;  /* (T,B,n)                                             */
    void *d_cov_smooth;   /* (T,B,n,n)                                           */
    void *d_mat_As;       /* (T+1,B,n,n)                                         */
    void *d_mat_Cs;       /* (T+1,B,p,n)                                         */
    void *d_Q, *d_R;      /* summed over T and B                                 */
    void *d_m0, *d_P0;    /* summed over B                                       */
    /* Alternative to g_obs/g_trans/g_post for the usual case that the terms only
     * enter an aggregated objective (kvae/kalman_vae.py:244-260): when g_scale is
     * not NULL the upstream gradient of every (t,b) entry is
     *   w_obs * g_scale[0], w_trans * g_scale[stride], w_post * g_scale[2*stride]
     * (g_scale: device pointer, same dtype; stride 0 = one scalar for all three) and
     * the three per-entry tensors are ignored. */
    const void *g_scale;
    int32_t g_scale_stride, _pad;
    double w_obs, w_trans, w_post;
} kvae_elbo_bwd;

int kvae_elbo_backward(const kvae_dims *d, const kvae_elbo_bwd *io,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* The Kalman part of the training objective in one call per direction: what
 * KalmanVariationalAutoencoder.elbo does between kvae/kalman_vae.py:141 and :260
 * (kalman_filter, kalman_smooth, zs_distrib.rsample(), the three log-density
 * terms, aggregate, the weighted sum) when no observation mask makes the mixing
 * weights depend on the filter state.  That call site keeps three scalars; the
 * twelve filter / smoother tensors and mat_As / mat_Cs are intermediate values
 * there, and these entry points never write them: per sequence-step the forward
 * call reads a, alpha, eps and writes only the state its backward call reads again.
 *
 * forward:  out (4) = [ w_obs * sum(obs), w_trans * sum(trans), w_post * sum(post),
 *   their total ]  with obs / trans / post as in kvae_elbo_forward, computed from
 *   the smoother's moments of the filter run on (a, alpha, mask); aggregate()'s
 *   constant (kvae/misc.py:21-41), kalman_weight and the minus sign of the
 *   posterior term are folded into the three weights.  `saved`:
 *   kvae_loss_saved_bytes(d) bytes the backward call needs unchanged (filtered,
 *   predicted and smoothed means and covariances).  terms: (3,T,B) per-pair values,
 *   or NULL.  mask: (T,B) or NULL; it enters the filter's blend only
 *   (kvae/state_space_model.py:313-325), the terms ignore it as in the reference.
 * backward: the cotangents of out[0..2] are g_scale[0], g_scale[stride],
 *   g_scale[2*stride] (device pointer; stride 0 = one scalar, the cotangent of the
 *   total); outputs d_a (T,B,p), d_alpha (T+1,B,K) and the parameter gradients summed
 *   over T and B (overwritten).  d->burn_in as in the separate calls.
 * Small state dimensions (those kvae_loss_supported() reports).  Deterministic: no
 * float atomics.
 * Three kernel forms serve these calls, chosen from the dimensions (KVAE_SEQ_LANES=1|4|16
 * in the environment forces one): one lane per sequence (large batches), four lanes per
 * sequence, and 16 lanes per sequence with the per-step state in LDS (float, n and p up to
 * 4, T <= 35, and no more sequences than are resident on the device at once: 4096 at
 * T = 20 on 256 compute units).  In the last form the forward call writes nothing to
 * `saved` (kvae_loss_saved_bytes() is then 256) and the backward call recomputes the
 * forward pass on chip instead of reading it back.                                 */
typedef struct {
    const void *a;        /* (T,B,p)                                             */
    const void *alpha;    /* (T+1,B,K)                                           */
    const void *mask;     /* (T,B) or NULL                                       */
    const void *A_K, *C_K, *Q, *R;
    const void *m0, *P0;  /* (n), (n,n): initial_state_mean / covariance         */
    const void *eps;      /* (T,B,n) standard-normal draw of zs_distrib.rsample()*/
    void *saved;          /* kvae_loss_saved_bytes(d)                            */
    void *terms;          /* (3,T,B) or NULL                                     */
    void *out;            /* (4)                                                 */
    double w_obs, w_trans, w_post;
    int32_t *status;
} kvae_loss_fwd;

typedef struct {
    kvae_loss_fwd fwd;    /* same pointers as the forward call                   */
    const void *g_scale;
    int32_t g_scale_stride, _pad;
    void *d_a, *d_alpha;
    void *d_A_K, *d_C_K, *d_Q, *d_R, *d_m0, *d_P0;   /* any may be NULL          */
} kvae_loss_bwd;

size_t kvae_loss_saved_bytes(const kvae_dims *d);
int kvae_loss_supported(int n, int p, int K, int dtype);
int kvae_kalman_loss_forward(const kvae_dims *d, const kvae_loss_fwd *io, void *workspace,
                             size_t workspace_bytes, void *stream);
int kvae_kalman_loss_backward(const kvae_dims *d, const kvae_loss_bwd *io, void *workspace,
                              size_t workspace_bytes, void *stream);

/* Both directions in one call, for the training step, which always differentiates the
 * objective it has just evaluated (examples/train.py:115-131 of the reference: elbo(...)
 * followed by (-elbo).backward()): everything kvae_kalman_loss_forward writes (io->fwd.out,
 * io->fwd.terms) and everything kvae_kalman_loss_backward writes, for the cotangents in
 * io->g_scale (NULL = 1 for each of the three totals, i.e. the gradient of out[3]; a caller
 * that learns the real cotangent later multiplies by it: the gradient is linear in it).
 * While a sequence's per-step state fits on chip (float, n and p up to 4, T <= 35, a modest
 * batch: 16 lanes per sequence, state in LDS) this is ONE launch that reads a, alpha, eps and
 * writes d_a, d_alpha and per-wavefront sums: no mean or covariance goes through device
 * memory and io->fwd.saved is not touched.  Otherwise it is the two calls above in a row
 * (io->fwd.saved and io->g_scale are then required).
 * Workspace: kvae_workspace_bytes(d, KVAE_OP_LOSS_VALUE_AND_GRAD).                  */
int kvae_kalman_loss_value_and_grad(const kvae_dims *d, const kvae_loss_bwd *io, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* One step of the masked filter's imputation, kvae/state_space_model.py:232-277, for
 * every sequence: z = A(w) m_pred (+ L_Q eps_z), a_imp = C(w) z (+ L_R eps_a),
 * a_w = mask * a + (1 - mask) * a_imp, with A(w) = sum_k w[b,k] A_K[k] (C alike).
 * Uses B, n, p, K, dtype of d.  w (B,K), m_pred (B,n), a (B,p), mask (B), eps_z (B,n) /
 * eps_a (B,p) standard-normal draws or NULL ("mean" mode), L_Q (n,n) / L_R (p,p) lower
 * Cholesky factors (needed with the draws), a_w (B,p) output: the observation the weight
 * network sees next.  Small state dimensions only (KVAE_E_UNSUPPORTED otherwise). */
int kvae_impute_observation(const kvae_dims *d, const void *w, const void *A_K, const void *C_K,
                            const void *m_pred, const void *a, const void *mask,
                            const void *eps_z, const void *L_Q, const void *eps_a,
                            const void *L_R, void *a_w, void *stream);

/* Adjoint of kvae_impute_observation for one time step (autograd through
 * kvae/state_space_model.py:232-277), given the mixed matrices A_t (B,n,n), C_t (B,p,n) the
 * filter step wrote.  The gradient w.r.t. a_w is d_aw_a + d_aw_b (d_aw_b may be NULL).
 * d_a (B,p): mask * that gradient is ADDED.  Written: d_aimp (B,p), the gradient w.r.t. the
 * imputed observation, and dz (B,n), w.r.t. the imputed state (their outer products with the
 * draws, summed over sequences and steps by the caller, are the gradients of L_R and L_Q);
 * gC (B,p,n) = d_aimp z^T and gA (B,n,n) = dz m_pred^T, the gradients w.r.t. C_t and A_t.
 * The gradient w.r.t. m_pred, A_t^T dz, joins what the one-step filter adjoint of this step
 * returned for its initial state (d_m_seg (B,n), d_P_seg (B,n,n)): with g_mn_prev (B,n) and
 * g_Pn_prev (B,n,n), the cotangents of the previous step's predicted state, d_m_seg + A_t^T dz
 * and d_P_seg are ADDED to them; with both NULL (first step) A_t^T dz is added to d_m_seg.
 * Uses B, n, p, dtype of d.  Small state dimensions only (KVAE_E_UNSUPPORTED otherwise). */
int kvae_impute_observation_backward(const kvae_dims *d, const void *d_aw_a, const void *d_aw_b,
                                     const void *mask, const void *A_t, const void *C_t,
                                     const void *m_pred, const void *eps_z, const void *L_Q,
                                     void *d_a, void *d_aimp, void *dz, void *gC, void *gA,
                                     void *d_m_seg, const void *d_P_seg, void *g_mn_prev,
                                     void *g_Pn_prev, void *stream);

/* The masked filter with its weight network evaluated inside the recursion, for the
 * two-layer perceptron softmax(W2 relu(W1 a + b1) + b2)
 * (kvae/dynamics_parameter_network.py:61-90 with num_layers = 2): the loop of
 * kvae/state_space_model.py:219-352 with an observation mask, in one launch.  Per step
 * the kernel imputes the observation the network sees (as kvae_impute_observation),
 * evaluates the network on it, and runs the filter step with the weights so obtained.
 * io as for kvae_filter_forward, except that io->alpha (T+1,B,K) is read only in row 0
 * (the weights of the first step, uniform in the reference) and rows 1..T are WRITTEN
 * (the caller owns the buffer; the const in the struct is not honoured here), and
 * io->mask NULL means nothing is masked.  Forward only (inference).  Small state
 * dimensions, hidden * (p + 1 + K) + K elements must fit 64 KB.                     */
typedef struct kvae_masked_mlp {
    int32_t hidden, _pad;
    const void *packed; /* hidden rows of [W1[j,0..p-1], b1[j], W2[0..K-1,j]], then b2 (K) */
    const void *eps_z;  /* (T,B,n) standard-normal draws or NULL ("mean" mode) */
    const void *L_Q;    /* (n,n) lower Cholesky factor of Q, needed with eps_z */
    const void *eps_a;  /* (T,B,p) or NULL */
    const void *L_R;    /* (p,p), needed with eps_a */
    void *a_w;          /* out (T,B,p): the observations the network saw */
} kvae_masked_mlp;

int kvae_filter_forward_masked_mlp(const kvae_dims *d, const kvae_filter_fwd *io,
                                   const kvae_masked_mlp *net, void *stream);

/* Open-loop rollout past the end of a filtered sequence: StateSpaceModel.predict_future,
 * kvae/state_space_model.py:486-575.  d->T is the number of NEW steps S written by the
 * call; B, n, p, K, dtype as usual.  Per new step, from the last one-step prediction
 * (mean_pred, cov_pred) and the last mixed observation matrix mat_C (= mat_Cs[T]):
 *   emit:    means <- mean_pred, covs <- cov_pred; z = mean_pred (+ chol(cov_pred) eps when
 *            `sample`, :526-533); a = mat_C z (:541)
 *   weights: w = net(a) (:544)
 *   advance: A = sum_k w_k A_K[k], C likewise (:545-548); mean_pred <- A mean_pred (:549);
 *            cov_pred <- sym(Q) when `sample`, else sym(A cov_pred A^T + Q) (:550-563)
 * phase KVAE_ROLLOUT_ALL: all S steps in one launch with the two-layer perceptron
 *   `packed_mlp` (layout of kvae_masked_mlp.packed, `hidden` units, must fit 64 KB)
 *   evaluated in the loop.
 * phase KVAE_ROLLOUT_EMIT / KVAE_ROLLOUT_ADVANCE (S must be 1): the two halves of one step
 *   for a caller that runs a recurrent weight network in between; ADVANCE reads the
 *   weights of the new step from `alpha` (B,K).
 * Values only.  Small state dimensions (KVAE_E_UNSUPPORTED otherwise).  A covariance that
 * is not positive definite when sampled sets bit 2 of *status.                         */
enum { KVAE_ROLLOUT_ALL = 0, KVAE_ROLLOUT_EMIT = 1, KVAE_ROLLOUT_ADVANCE = 2 };

typedef struct kvae_rollout {
    const void *A_K, *C_K, *Q;   /* (K,n,n), (K,p,n), (n,n) */
    const void *mean_pred;       /* (B,n)   m_{T|T-1}, the last row of the filter's mean_pred */
    const void *cov_pred;        /* (B,n,n) */
    const void *mat_C;           /* (B,p,n) the last row of mat_Cs; unused by ADVANCE */
    const void *eps;             /* (S,B,n) standard-normal draws; NULL unless `sample` */
    const void *packed_mlp;      /* ALL only */
    const void *alpha;           /* (B,K), ADVANCE only */
    int32_t hidden, phase, sample, _pad;
    void *as_new;                /* out (S,B,p)   the generated observations */
    void *means, *covs;          /* out (S,B,n), (S,B,n,n): rows appended to means / covariances */
    void *mean_pred_new;         /* out (S,B,n)   rows appended to next_means */
    void *cov_pred_new;          /* out (S,B,n,n) */
    void *mat_As_new;            /* out (S,B,n,n) rows appended to mat_As */
    void *mat_Cs_new;            /* out (S,B,p,n) */
    int32_t *status;
} kvae_rollout;

int kvae_predict_future(const kvae_dims *d, const kvae_rollout *io, void *stream);

/* Filter and smoother in one call, the sequence of kvae/kalman_vae.py:141-160
 * (and of the imputation path kvae/evaluation.py:100-140): the results of
 * kvae_filter_forward(d, f) followed by kvae_smooth_forward(d, s) on `stream`, every
 * tensor written (small state, 32768 sequences or more: one launch, see
 * kvae_filter_smooth_fused; otherwise the two launches).  The smoother must read what the filter wrote:
 * s->mean_filt == f->mean_filt, s->cov_filt == f->cov_filt, s->mean_pred ==
 * f->mean_pred, s->cov_pred == f->cov_pred, s->mat_As == f->mat_As, s->mat_Cs ==
 * f->mat_Cs (KVAE_E_BADARG otherwise).  There is no fused backward entry point: between
 * the two backward calls the caller's autograd adds the gradients of the other consumers
 * of mat_As / mat_Cs (the ELBO terms), so they stay two calls.                    */
int kvae_filter_smooth_forward(const kvae_dims *d, const kvae_filter_fwd *f,
                               const kvae_smooth_fwd *s, void *stream);

/* Adjoint of kvae_filter_forward_masked_mlp (kvae/state_space_model.py:219-352 under
 * autograd with an observation mask): `io` as for kvae_filter_backward, with io->fwd holding
 * the pointers of the forward call (alpha = the weights it wrote); `net->fwd` the forward
 * call's network arguments (a_w is read).  Per step the gradient of the weights of the next
 * step goes on through the softmax, the perceptron, the blend a_w = mask a + (1 - mask) a_imp
 * and the imputation a_imp = C_t (A_t m_pred + L_Q eps_z) + L_R eps_a (:232-277) into d_a,
 * the bases, the prediction it read, and the noise factors.
 *   g_a_w (T,B,p) / g_alpha (T+1,B,K): upstream gradients of the forward call's a_w / alpha
 *     outputs, or NULL.
 *   learn_weight_model 0: the weights were detached from the network (:281-282); only
 *     g_a_w flows into the imputation.
 *   d_logits (T,B,K): gradient w.r.t. the network's pre-softmax outputs of step t (which
 *     became alpha[t+1]); the caller obtains the gradients of the network's parameters by
 *     re-evaluating it on a_w and back-propagating d_logits.
 *   d_L_Q (n,n) / d_L_R (p,p): gradients w.r.t. the noise factors of the imputation, or NULL.
 * io->d_alpha is scratch here (the weights are internal).  Workspace:
 * kvae_filter_backward_masked_mlp_workspace_bytes(d).  Small state dimensions only.      */
typedef struct kvae_masked_mlp_bwd {
    kvae_masked_mlp fwd;
    const void *g_a_w, *g_alpha;
    void *d_logits, *d_L_Q, *d_L_R;
    int32_t learn_weight_model, _pad;
} kvae_masked_mlp_bwd;

size_t kvae_filter_backward_masked_mlp_workspace_bytes(const kvae_dims *d);
int kvae_filter_backward_masked_mlp(const kvae_dims *d, const kvae_filter_bwd *io,
                                    const kvae_masked_mlp_bwd *net, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* The same sequence (kvae/kalman_vae.py:141-160, with the weight network of masked
 * sequences of kvae/state_space_model.py:232-282 when `net` is given) as ONE launch that
 * writes only what the caller asks for: inference consumers such as the imputation sweeps
 * (kvae/evaluation.py:100-140) read `net->a_w` and `s->as_resampled` and nothing else.
 *   f: inputs as kvae_filter_forward; alpha (T+1,B,K) is read when net is NULL, otherwise
 *      row 0 is read and rows 1..T are written as by kvae_filter_forward_masked_mlp.  Every
 *      output of f (mean_filt, cov_filt, mean_pred, cov_pred, mat_As, mat_Cs) may be NULL.
 *   s: LR, eps_state, eps_obs as kvae_smooth_forward; its inputs mean_filt ... mat_Cs are
 *      ignored (the reverse sweep reads what the forward sweep of the same lane produced and
 *      re-mixes the per-step matrices from alpha); every output (mean_smooth, cov_smooth,
 *      zs, as_resampled) may be NULL.
 *   workspace: kvae_filter_smooth_fused_workspace_bytes(d, f) bytes, room for those of
 *      mean_filt / cov_filt that are NULL in f (may be NULL when 0); the one-step predictions
 *      are recomputed by the reverse sweep, not read back.
 * Forward only.  Small state dimensions (KVAE_E_UNSUPPORTED otherwise: use the two calls). */
size_t kvae_filter_smooth_fused_workspace_bytes(const kvae_dims *d, const kvae_filter_fwd *f);
int kvae_filter_smooth_fused(const kvae_dims *d, const kvae_filter_fwd *f,
                             const kvae_masked_mlp *net, const kvae_smooth_fwd *s,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------ */
/* Recurrent part of the LSTM dynamics-parameter network, one layer over a whole
 * sequence: kvae/dynamics_parameter_network.py:46-58 (LSTMModel.forward, i.e.
 * torch.nn.LSTM with gate order i,f,g,o and zero initial state), float32 only,
 * hidden size H in {32, 64, 128, 256}.  Used when the mixing weights of all
 * steps can be produced up front (no observation mask).
 *
 * forward: gates (T,B,4H) holds on entry the input projection
 *   X_t W_ih^T + b_ih + b_hh of every step (one time-batched library GEMM done
 *   by the caller) and on return the activated gates i,f,g,o; WhhT is W_hh
 *   transposed, (H,4H); h, c: (T,B,H) outputs.
 * backward: dh (T,B,H) holds on entry the gradient w.r.t. every h_t and is used
 *   as the accumulator of the recurrent gradient (overwritten); dgates (T,B,4H)
 *   receives the gradient w.r.t. the gate pre-activations, from which the caller
 *   forms dW_ih, dW_hh, db and dX with time-batched library GEMMs; Whh is
 *   (4H,H); dc_work: (B,H) scratch.
 */
int kvae_lstm_layer_forward(int T, int B, int H, void *gates, const void *WhhT, void *h, void *c,
                            void *stream);
int kvae_lstm_layer_backward(int T, int B, int H, const void *gates, const void *c, void *dh,
                             const void *Whh, void *dgates, void *dc_work, void *stream);

/* Persistent form for the reference's default network (2 layers, H = 128,
 * layer-0 input dimension P <= 4): ONE launch runs all T steps of both layers
 * for a tile of 16 or 32 sequences (chosen by batch size), h_t in LDS, c_t in
 * registers, weights streamed from L2.
 * forward: x (T,B,P), Wih0 (4H,P), b0 = b_ih0 + b_hh0 (4H), Whh0T / Wih1T /
 *   Whh1T: the (4H,H) matrices transposed to (H,4H), b1 = b_ih1 + b_hh1; outputs
 *   gates0/1 (T,B,4H) activated (both may be null when no backward will follow),
 *   h0/1 and c0/1 (T,B,H).  h_init / c_init: (2,B,H) state to start from, both null
 *   for the zero state - the cached state LSTMModel carries between calls
 *   (dynamics_parameter_network.py:28-33, 53), used by the step-by-step masked filter.
 * backward: g_h1 (T,B,H) gradient w.r.t. the top layer's h; Wih0 (4H,P); Whh0,
 *   Wih1, Whh1 as stored (4H,H).  Outputs: dG0, dG1 (T,B,4H), the gate
 *   pre-activation gradients (operands of the caller's three time-batched
 *   recurrent/input weight-gradient GEMMs); dx (T,B,P); and
 *   partial (kvae_lstm2_backward_blocks(B), 4H*(4+P)): per-workgroup sums whose
 *   column sum over the first axis is [ db0 | db0 | db1 | db1 (4H each) | dWih0 (4H,P) ]
 *   (each bias sum twice: one copy for bias_ih, one for bias_hh of torch.nn.LSTM).
 */
int kvae_lstm2_forward(int T, int B, int P, const void *x, const void *Wih0, const void *b0,
                       const void *Whh0T, const void *Wih1T, const void *Whh1T, const void *b1,
                       void *gates0, void *h0, void *c0, void *gates1, void *h1, void *c1,
                       const void *h_init, const void *c_init, void *stream);
int kvae_lstm2_backward_blocks(int B);
/* The forward kernel's weight operands from torch.nn.LSTM's parameters
 * (weight_hh_l0, weight_ih_l1, weight_hh_l1: (4H,H); the four bias vectors) in one
 * launch: out (kvae_lstm2_prepared_elems() floats) = [Whh0T | Wih1T | Whh1T | b0 | b1]. */
long kvae_lstm2_prepared_elems(void);
int kvae_lstm2_prepare(const void *w_hh0, const void *w_ih1, const void *w_hh1, const void *b_ih0,
                       const void *b_hh0, const void *b_ih1, const void *b_hh1, void *out,
                       void *stream);
int kvae_lstm2_backward(int T, int B, int P, const void *g_h1, const void *x, const void *gates0,
                        const void *c0, const void *gates1, const void *c1, const void *Wih0,
                        const void *Whh0, const void *Wih1, const void *Whh1, void *dG0, void *dG1,
                        void *dx, void *partial, void *stream);

/* One layer of that network's LSTM (hidden size 128) over a whole sequence in ONE launch
 * per direction, with the layer's 128 x 512 recurrent matrix held in registers for all T
 * steps (nothing is streamed from L2 inside the time loop).  Same function as one layer of
 * torch.nn.LSTM, kvae/dynamics_parameter_network.py:46-58.  float32.
 * forward: P in 1..4: the layer input x (T,B,P) is projected in the kernel with Wih (4H,P)
 *   and bias (4H) = b_ih + b_hh; P == 0: gates (T,B,4H) holds the input projection (bias
 *   included, one time-batched library GEMM by the caller) on entry.  WhhT: W_hh transposed,
 *   (H,4H).  store_gates != 0: the activated gates i,f,g,o of every step are written to
 *   gates (in place for P == 0); h, c: (T,B,H) outputs.  h_init / c_init: (B,H) state to
 *   start from, both NULL for the zero state.
 * backward: g_h (T,B,H) gradient w.r.t. every h_t from above; gates, c as the forward call
 *   left them; c_init as in the forward call; Whh (4H,H) as stored.  Outputs: dG (T,B,4H),
 *   the gate pre-activation gradients (operands of the caller's time-batched weight-gradient
 *   GEMMs and of the product that carries the gradient to the layer below);
 *   partial (kvae_lstm128_blocks(B), 4H*(2+P)): per-workgroup sums whose column sum over the
 *   first axis is [ db | db (4H each: bias_ih and bias_hh) | dWih (4H,P) ]; for P > 0 also
 *   dx (T,B,P) = dG Wih.                                                                */
int kvae_lstm128_blocks(int B);
/* How the two kernels above form their matrix products.  split = 1 (the default): every
 * fp32 product a*b is a_hi*b_hi + a_lo*b_hi + a_hi*b_lo on the bf16 matrix instruction with
 * fp32 accumulation (x = x_hi + x_lo, both bf16): about 1e-5 relative per product (5e-5 at worst), 1e-6 on
 * a 128-term gate sum, a fifth of the matrix-instruction time.  split = 0: the fp32 matrix
 * instruction, exact fp32 products.  KVAE_LSTM_PRECISION=fp32 in the environment selects 0
 * at the first call.  Process-wide; no counterpart in the reference (torch.nn.LSTM's
 * precision is the library's).                                                          */
int kvae_lstm128_set_precision(int split);
int kvae_lstm128_get_precision(void);
int kvae_lstm128_forward(int T, int B, int P, const void *x, const void *Wih, const void *bias,
                         const void *WhhT, void *gates, void *h, void *c, const void *h_init,
                         const void *c_init, int store_gates, void *stream);
int kvae_lstm128_backward(int T, int B, int P, const void *x, const void *Wih, const void *gates,
                          const void *c, const void *c_init, const void *g_h, const void *Whh,
                          void *dG, void *dx, void *partial, void *stream);

/* kvae_lstm128_backward for the upper layer (P = 0) with the network's softmax head
 * alpha = softmax(h W^T + b), W (K,H), K in 2..4, on top of it
 * (kvae/dynamics_parameter_network.py:57-58): instead of reading the gradient w.r.t. h the
 * kernel forms it from alpha (T,B,K) and g_alpha (T,B,K), the gradient w.r.t. alpha, and also
 * sums the head's parameter gradients: head_partial (kvae_lstm128_blocks(B), K*(H+1)) whose
 * column sum over the first axis is [ dW (K,H) | db (K) ] (the layout of
 * kvae_softmax_head_backward).  Everything else as kvae_lstm128_backward.  Other K:
 * KVAE_E_UNSUPPORTED (use kvae_softmax_head_backward, then kvae_lstm128_backward).      */
int kvae_lstm128_backward_head(int T, int B, int K, const void *gates, const void *c,
                               const void *c_init, const void *alpha, const void *g_alpha,
                               const void *Wh, const void *Whh, void *dG, void *partial,
                               void *head_partial, void *stream);

/* ONE time step of one layer of that LSTM (hidden size 128) over B rows, input projection
 * included: gates = x W_ih^T + h_prev W_hh^T + bias, c = f c_prev + i g, h = o tanh(c) -- one
 * step of torch.nn.LSTM as the masked filter calls the network once per time step
 * (kvae/state_space_model.py:279-282, kvae/dynamics_parameter_network.py:46-58).
 * P == 0, the upper layer: x (B,128), Wx = W_ih transposed (128,512); each workgroup keeps
 *   the slice of both matrices that belongs to 64 hidden units in registers, so the (B,512)
 *   projection never goes through memory (3.5 KB per row instead of 6.5 KB for
 *   kvae_rowstream_matmul + kvae_lstm128_forward).
 * P in 1..2, the lower layer: x (B,P), Wx = W_ih as stored (512,P), projected with plain
 *   multiply-adds (2 KB per row); wider inputs: KVAE_E_UNSUPPORTED (kvae_lstm128_forward).
 * h_prev, c_prev, h_out, c_out (B,128); h_prev and c_prev both NULL for the zero state;
 * WhhT (128,512) = W_hh transposed; bias (512) = b_ih + b_hh.  float32; h_prev (and x for
 * P == 0) 16-byte aligned.  Several tiles of rows are kept in flight per compute unit.
 * Split product form.                                                                   */
int kvae_lstm128_step_fused(long B, int P, const void *x, const void *Wx, const void *h_prev,
                            const void *c_prev, const void *WhhT, const void *bias, void *h_out,
                            void *c_out, void *stream);

/* ONE time step of one layer's adjoint, without its matrix products: the masked filter calls
 * the network once per time step (kvae/state_space_model.py:279-282), so the adjoint of
 * torch.nn.LSTM (kvae/dynamics_parameter_network.py:46-58) is walked one step at a time, the
 * state gradients carried by the caller.  gates (B,512): the activated i,f,g,o of the step as
 * kvae_lstm128_forward stored them; c_prev (B,128) or NULL (zero state), c (B,128); the
 * gradient w.r.t. this step's h is dh_a + dh_b (dh_b may be NULL: what comes from above and
 * the recurrent part dG_{t+1} W_hh); dc_in (B,128) or NULL: the cell-state gradient from step
 * t+1.  Outputs: dG (B,512), the gate pre-activation gradients (operands of
 * kvae_rowstream_matmul for the gradients w.r.t. h_{t-1} and the layer input, and of
 * kvae_lstm128_wgrad), and dc_prev (B,128).  dx (B,P) or NULL: for a layer with a narrow
 * input (P <= 8) the gradient w.r.t. that input, dG Wih with Wih (512,P) as stored, plus
 * dx_add (B,P) when that is given.  float32; the (B,128) and (B,512) operands 16-byte
 * aligned.                                                                              */
int kvae_lstm128_cell_backward(long B, const void *gates, const void *c_prev, const void *c,
                               const void *dh_a, const void *dh_b, const void *dc_in, void *dG,
                               void *dc_prev, const void *Wih, int P, const void *dx_add,
                               void *dx, void *stream);

/* Y = X W (+ bias), the two time-batched products of that network's LSTM outside the
 * recurrences: layer 1's input projection of all steps (K = 128, N = 512, W = weight_ih_l1
 * transposed, bias = b_ih + b_hh) and the gradient layer 1 hands to layer 0 (K = 512,
 * N = 128, W = weight_ih_l1 as stored, bias NULL) -- torch.nn.LSTM and its autograd inside
 * kvae/dynamics_parameter_network.py:46-58.  X (M,K), W (K,N), bias (N) or NULL, Y (M,N),
 * float32 row-major, 16-byte aligned; any M >= 1.  W stays in registers, X and Y stream
 * through once; products are formed as in kvae_lstm128_forward with split = 1.  Other
 * (K,N): KVAE_E_UNSUPPORTED.                                                              */
int kvae_rowstream_matmul(long M, int K, int N, const void *X, const void *W, const void *bias,
                          void *Y, void *stream);

/* Weight gradient of one such LSTM layer: dW = dG^T H with dG (R,512) the gate gradients
 * kvae_lstm128_backward wrote and H (R,128) the layer's input or its own h one step earlier
 * (autograd of torch.nn.LSTM, kvae/dynamics_parameter_network.py:46-58); R = T*B.
 * Row m of dG is paired with row m - shift of H (rows m < shift contribute nothing):
 * shift = 0 for weight_ih, shift = B for weight_hh.  The rows are cut into
 * kvae_lstm128_wgrad_slabs(R) slabs; slab s writes its (512,128) float32 result at
 * partial + s * partial_stride (elements, >= 65536), and the sum over s (kvae_column_sum)
 * is the gradient: fixed summation order, no atomics.  dG, H 16-byte aligned.  Products are
 * formed as in kvae_lstm128_forward with split = 1.                                       */
int kvae_lstm128_wgrad_slabs(long R);
int kvae_lstm128_wgrad(long R, const void *dG, const void *H, long shift, void *partial,
                       long partial_stride, void *stream);

/* Softmax head of the dynamics-parameter network, alpha = softmax(h W^T + b):
 * kvae/dynamics_parameter_network.py:57-58 (LSTMModel) and 84-90 (last layer of
 * MLPModel).  float32; h (R,H) with R = T*B rows and H in {32,64,128,256};
 * W (K,H), b (K), K <= 8; alpha (lead_rows + R, K): the first lead_rows rows are
 * filled with 1/K (the weights the filter uses before it has seen an observation,
 * kvae/state_space_model.py:225: pass lead_rows = B to get the (T+1,B,K) weight
 * tensor in one piece), the softmax rows follow.
 * backward: g_alpha (R,K) gradient w.r.t. alpha; outputs d_h (R,H) and
 *   partial (kvae_softmax_head_backward_blocks(R), K*(H+1)) whose column sum
 *   over the first axis is [ dW (K,H) | db (K) ].
 */
int kvae_softmax_head_forward(long R, int H, int K, const void *h, const void *W, const void *b,
                              void *alpha, long lead_rows, void *stream);
int kvae_softmax_head_backward_blocks(long R);
int kvae_softmax_head_backward(long R, int H, int K, const void *h, const void *W,
                               const void *alpha, const void *g_alpha, void *d_h, void *partial,
                               void *stream);

/* Cholesky factor L (lower, zeros above) of one small SPD matrix A (n,n) and its
 * adjoint: the factor torch.distributions.MultivariateNormal builds for the
 * noise covariances in kvae/state_space_model.py:242-247, 261-266, 409-413.
 * n <= 64 (float) / 32 (double).  No host synchronisation and no allocation
 * (usable under HIP graph capture): a non-positive pivot sets bit 8 of *status.
 * backward: gL (n,n) gradient w.r.t. L (lower triangle used); gA symmetric.
 */
int kvae_cholesky_forward(int n, int dtype, const void *A, void *L, void *status, void *stream);
int kvae_cholesky_backward(int n, int dtype, const void *L, const void *gL, void *gA, void *stream);

/* The noise covariances and the factor of R in one launch:
 *   Q (n,n) = Lq Lq^T + q_reg I,  R (p,p) = Lr Lr^T + r_reg I   (the mat_Q / mat_R
 *   properties, kvae/state_space_model.py:112-146),  LR (p,p) = chol(R) (null = skip).
 * Forward values only (for steps that do not learn the noise factors).  Same limits
 * and failure reporting as kvae_cholesky_forward.
 */
int kvae_noise_factors(int n, int p, int dtype, const void *Lq, const void *Lr, double q_reg,
                       double r_reg, void *Q, void *R, void *LR, void *status, void *stream);

/* Deterministic tree sums (no atomics, no memset; same bits on every run).
 * kvae_column_sum: out (cols) = column sums of the row-major (rows, cols) matrix
 *   `in` - used for the per-workgroup partial results of the backward kernels.
 * kvae_vector_sum: out (batch) = total of each of `batch` contiguous vectors of
 *   length n - the (T,B) -> scalar reduction of the ELBO terms, kvae/misc.py:21-41
 *   (`aggregate` is this sum times a constant).  scratch: at least
 *   kvae_vector_sum_scratch_elems(batch, n) elements of the same dtype (may be 0).
 */
int kvae_column_sum(long rows, int cols, int dtype, const void *in, void *out, void *stream);
/* The column sums of up to four matrices of one dtype in ONE launch: rows[i], cols[i], in[i],
 * out[i] as in kvae_column_sum (host arrays of `parts` entries, read before the call returns). */
int kvae_column_sum_many(int parts, const long *rows, const int *cols, int dtype,
                         const void *const *in, void *const *out, void *stream);
long kvae_vector_sum_scratch_elems(int batch, long n);
int kvae_vector_sum(int batch, long n, int dtype, const void *in, void *scratch, void *out,
                    void *stream);

/* The three ELBO terms to the objective in two launches: terms (3,n) holds obs,
 * trans, post (n = T*B entries each); out (4) = [ w_obs * sum(obs), w_trans *
 * sum(trans), w_post * sum(post), their total ]  - aggregate() of kvae/misc.py:21-41
 * with its constant folded into the weights, and kalman_vae.py:244-260.  scratch:
 * kvae_weighted_terms_scratch_elems(n) elements.  Deterministic. */
long kvae_weighted_terms_scratch_elems(long n);
int kvae_weighted_terms(long n, int dtype, const void *terms, void *scratch, double w_obs,
                        double w_trans, double w_post, void *out, void *stream);

/* Number of memset / kernel nodes in a captured hipGraph_t (child graphs
 * included), negative on error.  Host-only helpers with no counterpart in the
 * reference: the graph-replay wrapper refuses graphs with memset nodes, whose
 * ordering against dependent kernels is unreliable on ROCm 7.0.              */
int kvae_graph_count_memset_nodes(void *hip_graph);
int kvae_graph_count_kernel_nodes(void *hip_graph);

/* ------------------------------------------------------------------------ */
enum {
    KVAE_OP_FILTER_BWD = 1, KVAE_OP_SMOOTH_BWD = 2, KVAE_OP_ELBO_BWD = 3,
    KVAE_OP_LOSS_FWD = 4, KVAE_OP_LOSS_BWD = 5, KVAE_OP_LOSS_VALUE_AND_GRAD = 6
};

/* Scratch bytes an entry point with a workspace argument needs (per-wave partial
 * sums; for KVAE_OP_LOSS_BWD also the gradients handed from its first to its second
 * sweep); 0 for an unknown op.                                                  */
size_t kvae_workspace_bytes(const kvae_dims *d, int op);

/* 1 when a kernel for (n,p,K,dtype) is compiled in.                          */
int kvae_supported(int n, int p, int K, int dtype);

int kvae_version(void);
const char *kvae_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* KVAE_KALMAN_H */
