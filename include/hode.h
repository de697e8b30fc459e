/*
 * hode.h -- C ABI of libhode.so: MI355X (gfx950) kernels for the hybrid-ODE hot path.
 *
 * The reference (ZhaozhiQIAN/Hybrid-ODE-NeurIPS-2021) has no FFI layer of its own; the operator
 * boundary of the path is two Python call sites, and every entry point below replaces one of them:
 *
 *   torchdiffeq.odeint(func, y0, t, rtol, atol, method, options)   reference model.py:1116, :837, :842
 *       func = RocheODE.forward      model.py:515-555   (HODE_RHS_ROCHE / HODE_RHS_ROCHE_ABLATE)
 *       func = NeuralODE.forward     model.py:1019-1026 (HODE_RHS_NEURAL)
 *       func = RocheODEReal.forward  model.py:613-645   (HODE_RHS_ROCHE_REAL)
 *     -> hode_rk_fwd / hode_rk_bwd          (method in {"euler","midpoint","rk4"})
 *     -> hode_dopri5_fwd / hode_dopri5_bwd  (method "dopri5", the reference default, sim_config.py:50)
 *     The backward entry points replace autograd's replay of the solver ops triggered by
 *     loss.backward() at reference training_utils.py:50 (the reference imports plain odeint, model.py:9-10,
 *     so its gradient is the discrete adjoint of the RK scheme, which is what *_bwd compute).
 *
 *   nn.LSTM(input_dim, hidden_dim)(obs, hidden) stepped over the window        reference model.py:420-422
 *     -> hode_lstm_fwd / hode_lstm_bwd      (masked, reverse- or forward-time single-layer LSTM)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer is a DEVICE pointer owned by the caller
 *     unless named host_*.  All floating-point buffers are fp32 (reference global_config.py:3).
 *   - time-major layouts exactly as the reference holds them: h[T][B][D], action/x/mask [T][B][.]
 *   - no allocation, no stream synchronisation and no global mutable state inside the library (the only
 *     exception: hode_dopri5_fwd reads back one 32-byte controller record per chunk of attempts, because
 *     the number of adaptive steps is data dependent; it says so below).  One call = launches on
 *     exactly the given stream.  Calls are thread-safe.
 *   - return value: 0 ok; <0 argument error (HODE_E_*); >0 a hipError_t from a launch.
 *     hode_last_error_string() returns a thread-local description of the last non-zero return.
 *   - numerical failure (non-finite state, dt underflow) is reported through desc->status, a device int32
 *     word the caller zeroes: bit 0 non-finite state, bit 1 dt underflow, bit 2 max_steps exceeded.
 *     The Python wrapper turns it into RuntimeError so that reference training_utils.py:43-47 keeps working.
 */
#ifndef HODE_H_
#define HODE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HODE_ABI_VERSION 1

/* right-hand side kinds */
#define HODE_RHS_ROCHE 0        /* expert PK/PD block + tanh(W y + b)          model.py:515-555 */
#define HODE_RHS_ROCHE_ABLATE 1 /* linear-oscillator expert block (ablate=True) model.py:545-549 */
#define HODE_RHS_NEURAL 2       /* tanh(W2 tanh(W1 [y,Dose] + b1) + b2)          model.py:1019-1026 */
#define HODE_RHS_ROCHE_REAL 3   /* two small MLPs + GRU-ODE block                model.py:613-645 */

/* fixed-grid methods (torchdiffeq names "euler", "midpoint", "rk4" = 3/8 rule) */
#define HODE_METHOD_EULER 0
#define HODE_METHOD_MIDPOINT 1
#define HODE_METHOD_RK4_38 2

/* descriptor flags */
#define HODE_FLAG_SKIP_FOLD 1 /* backward: leave the per-wave gradient partials unfolded (diagnostics: lets a caller time
                                the adjoint kernel alone; grad_w1 / grad_b1 / grad_theta are then NOT updated) */

#define HODE_FLAG_OVERWRITE_GRADS 2 /* backward: STORE grad_w1 / grad_b1 / grad_theta instead of accumulating into them (the
                                      caller then need not zero them).  hode_rk_bwd with the ROCHE kinds only; HODE_E_UNSUPPORTED elsewhere */

#define HODE_FLAG_TAPE 4 /* fixed-grid ROCHE solves: hode_rk_fwd leaves the expert block's intermediate stage states in
                            `workspace` and hode_rk_bwd reads them back instead of re-integrating every step.  Set it in
                            BOTH calls, size the buffer with hode_workspace_bytes(d, HODE_WS_RK_FWD / _BWD) with the flag
                            set, and hand the same, untouched buffer to both.  Costs 16 (stages-1) B per patient and grid
                            interval of extra HBM traffic in each direction, plus, for rk4, 8 (D-4) B for the learned
                            block's last two stage derivatives; layouts without a tape ignore the flag.  With the tape
                            the adjoint kernel runs 5 waves per 48 patients (the fifth forms the learned block's contribution
                            to the expert cotangent), 6 with need_theta_grad (one more accumulates the expert-parameter
                            gradients); every output is bit-identical to the tape-less path. */

#define HODE_FLAG_DETACH_FIRST_STEP 8 /* hode_dopri5_bwd: treat the first step size as a constant.  torchdiffeq computes
                                        Hairer's dt_0 from y0, f0, f1 with autograd ON, so the reference's loss.backward()
                                        differentiates it (every later dt is a controller constant); the default follows the
                                        reference, this flag drops that term (diagnostics / tests that separate the two) */

#define HODE_FLAG_NO_TAPE 16 /* hode_dopri5_fwd: no backward will follow (evaluation).  The accepted-state tape -- otherwise
                               (max_steps + 1) * B * D * 4 bytes -- shrinks to two rows; max_steps then only bounds the
                               24-byte-per-step time records.  hode_dopri5_bwd refuses a descriptor with this flag. */

/* argument errors */
#define HODE_E_NULL -1      /* a required pointer is NULL */
#define HODE_E_SIZE -2      /* struct_size mismatch / non-positive dimension */
#define HODE_E_UNSUPPORTED -3 /* (rhs_kind, latent_dim, method) combination has no kernel */
#define HODE_E_WORKSPACE -4 /* workspace too small (see hode_workspace_bytes) */
#define HODE_E_ALIGN -5     /* pointer not 16-byte aligned where the kernel needs it */

/* status word bits (device int32, OR-ed by the kernels) */
#define HODE_STATUS_NONFINITE 1
#define HODE_STATUS_DT_UNDERFLOW 2
#define HODE_STATUS_MAX_STEPS 4

/* layout of theta for HODE_RHS_ROCHE: creation order of the 13 scalars at reference model.py:468-482 */
enum {
  HODE_TH_HILL_CURE = 0, HODE_TH_HILL_PATHO, HODE_TH_EC50_PATHO, HODE_TH_EMAX_PATHO, HODE_TH_K_DEXA,
  HODE_TH_K_DISCURE_IMMUNEREACT, HODE_TH_K_DISCURE_IMMUNITY, HODE_TH_K_DISPROG, HODE_TH_K_IMMUNE_DISEASE,
  HODE_TH_K_IMMUNE_FEEDBACK, HODE_TH_K_IMMUNE_OFF, HODE_TH_K_IMMUNITY, HODE_TH_KEL,
  HODE_TH_THETA_1, HODE_TH_THETA_2, /* ablate only, model.py:484-485 */
  HODE_N_THETA = 16
};

/*
 * One descriptor for every solver entry point (fixed-grid and adaptive, forward and backward).
 * Unused pointers may be NULL.  "acc" = accumulator the caller zero-fills; the kernels add into it.
 */
typedef struct hode_solve_desc {
  uint32_t struct_size;   /* = sizeof(hode_solve_desc) */
  int32_t rhs_kind;       /* HODE_RHS_* */
  int32_t method;         /* HODE_METHOD_* (ignored by dopri5 entry points) */
  int32_t perturb;        /* torchdiffeq options["perturb"] (only DecoderReal sets it, model.py:826) */
  int32_t batch;          /* B patients */
  int32_t latent_dim;     /* D = 4 expert + (D-4) learned */
  int32_t n_times;        /* T = len(t) */
  int32_t n_dose;         /* K doses per patient (ROCHE / NEURAL), equal for all patients (model.py:507) */
  int32_t hidden_dim;     /* NEURAL: 10*D; ROCHE_REAL: MLP hidden width */
  int32_t n_action_times; /* ROCHE_REAL: length of the dose table along time */
  int32_t lanes_per_patient; /* 0 = library chooses; forces a variant (tuning / tests): 1 = lane per patient,
                                4 = patient per DPP quad, 16 = MFMA layout (4 lanes strided by 16, D in {8,12,16}),
                                48 = split layout (expert wave + learned waves per 48 patients, D in {8,12}; default there) */
  int32_t need_theta_grad;   /* backward: also accumulate grad_theta */

  const float* t;          /* [T] output grid == step grid (model.py:1072); strictly increasing */
  const float* y0;         /* [B][D] */
  const float* dosage;     /* ROCHE/NEURAL: [B] (set_action, model.py:498); ROCHE_REAL: [Ta][B] */
  const float* dose_times; /* ROCHE/NEURAL: [B][K] fp32 times (model.py:502-507) */
  const float* theta;      /* ROCHE: [HODE_N_THETA]; ROCHE_REAL: {k_immunity, kel, kel2} */
  const float* w1;         /* ROCHE: ml_net.0.weight [D-4][D]; NEURAL: [10D][D+1]; REAL: all weights flat in parameter
                              creation order (model.py:588-607): dx1_net.0.{w[H][3],b[H]}, dx1_net.2.{w[H],b[1]},
                              dx2_net.0.{w[H][2],b[H]}, dx2_net.2.{w[H],b[1]}, lin_hh, lin_hz, lin_hr [D-4][D-4] */
  const float* b1;         /* ROCHE: ml_net.0.bias [D-4];      NEURAL: [10D] */
  const float* w2;         /* NEURAL: [D][10D] */
  const float* b2;         /* NEURAL: [D] */
  float* h;                /* forward out / backward in: [T][B][D], h[0] = y0 */
  int32_t* status;         /* device status word (may be NULL for fixed-grid calls) */

  /* backward */
  const float* grad_h;     /* [T][B][D] cotangent of h */
  float* grad_y0;          /* out [B][D] */
  float* grad_w1;          /* acc, shape of w1 */
  float* grad_b1;          /* acc */
  float* grad_w2;          /* acc */
  float* grad_b2;          /* acc */
  float* grad_theta;       /* acc [HODE_N_THETA] */

  /* adaptive (dopri5) */
  double rtol, atol;       /* reference passes 1e-7 / 1e-8 (model.py:1079-1080) */
  int32_t max_steps;       /* tape capacity in accepted steps */
  int32_t flags;           /* HODE_FLAG_* (0 in normal use) */
  int32_t* host_n_accepted; /* HOST out (dopri5_fwd): accepted steps written to the tape */
  int32_t* host_n_rejected; /* HOST out (dopri5_fwd) */

  void* workspace;         /* device scratch, >= hode_workspace_bytes(desc, which) */
  size_t workspace_bytes;
} hode_solve_desc;

/* LSTM encoder (reference EncoderLSTM.forward model.py:408-428, EncoderLSTMReal.forward :210-242).
 * The kernel fuses the reference's cat([x, a]) * cat([mask, 1]) (model.py:415-421): x and mask cover the first
 * obs_dim input columns, a the remaining input_dim - obs_dim (never masked). */
typedef struct hode_lstm_desc {
  uint32_t struct_size;
  int32_t seq_len;     /* T */
  int32_t batch;       /* B */
  int32_t input_dim;   /* I = obs_dim + action columns */
  int32_t hidden_dim;  /* H */
  int32_t obs_dim;     /* columns taken from x (and masked); the other I - obs_dim come from a */
  int32_t reverse;     /* 1: walk t = T-1 .. 0 (EncoderLSTM, model.py:420); 0: forward (EncoderLSTMReal) */
  int32_t save_tape;   /* forward: 1 = write the gate/cell tape into workspace for hode_lstm_bwd */
  const float* x;      /* [T][B][obs_dim] */
  const float* a;      /* [T][B][I - obs_dim] or NULL when I == obs_dim */
  const float* mask;   /* [T][B][obs_dim] or NULL (no masking) */
  const float* w_ih;   /* [4H][I]  gate order i,f,g,o (nn.LSTM) */
  const float* w_hh;   /* [4H][H] */
  const float* b_ih;   /* [4H] */
  const float* b_hh;   /* [4H] */
  float* h_out;        /* [B][H] final hidden state */
  float* c_out;        /* [B][H] final cell state */
  /* backward */
  const float* grad_h_out; /* [B][H] cotangent of h_out */
  float* grad_gates;   /* out [T][B][4H]: d loss / d pre-activation gates per step (feeds the weight-gradient GEMMs) */
  float* h_prev;       /* out: the weight-gradient GEMM operand [T][B][W], W = roundup4(I + H + 1), per (t, b) row:
                          obs_dim columns LEFT UNTOUCHED for the caller's x*mask | action columns (I - obs_dim) | hidden
                          state entering the step (H) | 1 | 0...   grad_gates^T h_prev over K = T*B then is
                          [grad_w_ih | grad_w_hh | grad_b_ih = grad_b_hh | 0] */
  void* workspace;     /* >= hode_lstm_workspace_bytes: packed weights (+ tape when save_tape) */
  size_t workspace_bytes;
} hode_lstm_desc;

/* Fused readout + masked SSE (reference model.py:1120 x_hat = Linear(D -> obs)(h) and model.py:1179
 * lik = sum((x - x_hat)^2 * mask) / B, plus their autograd backward) without materialising x_hat. */
typedef struct hode_readout_desc {
  uint32_t struct_size;
  int32_t latent_dim;   /* D */
  int32_t obs_dim;      /* obs (multiple of 4, <= 128) */
  float scale;          /* 1 / B: folded into the gradients (lik itself is returned as the plain sum) */
  int64_t rows;         /* T * B rows of h / x / mask */
  const float* h;       /* [rows][D] */
  const float* x;       /* [rows][obs] */
  const float* mask;    /* [rows][obs] */
  const float* w;       /* output_function.0.weight [obs][D] */
  const float* b;       /* output_function.0.bias [obs] */
  float* lik;           /* out [1]: sum((x - x_hat)^2 * mask) */
  float* grad_h;        /* out [rows][D] or NULL for the loss only: d(scale * lik)/dh */
  float* grad_w;        /* acc [obs][D]: d(scale * lik)/dW */
  float* grad_b;        /* acc [obs] */
  void* workspace;      /* >= hode_readout_workspace_bytes */
  size_t workspace_bytes;
} hode_readout_desc;

/* Fused two-layer readout + masked, optionally time-weighted SSE of the real-data model: x_hat = W2 ELU(W1 h + b1) + b2
 * (DecoderReal.output_function, reference model.py:809-813, applied at :859) and
 * lik = sum((x - x_hat)^2 * mask * weight[t]) (VariationalInferenceReal.loss, model.py:1243-1247), plus every gradient,
 * in one pass over the rows on the matrix cores; x_hat is never written.  Compiled for latent_dim in {20, 4},
 * hidden_dim = latent_dim + 1, obs_dim = 24 (the DDW schema). */
typedef struct hode_readout_mlp_desc {
  uint32_t struct_size;
  int32_t latent_dim;    /* D */
  int32_t hidden_dim;    /* D + 1 */
  int32_t obs_dim;       /* 24 */
  int32_t batch;         /* B: rows are (t, b) time-major, t = row / B indexes time_weight */
  float scale;           /* 1 / B: folded into the gradients (lik itself is returned as the plain weighted sum) */
  int64_t rows;          /* T' * B */
  const float* h;        /* [rows][D] */
  const float* x;        /* [rows][obs] */
  const float* mask;     /* [rows][obs] */
  const float* time_weight; /* [T'] or NULL (weight = 1; model.py:1243-1246) */
  const float* w1;       /* output_function.0.weight [D+1][D] */
  const float* b1;       /* [D+1] */
  const float* w2;       /* output_function.2.weight [obs][D+1] */
  const float* b2;       /* [obs] */
  float* lik;            /* out [1] */
  float* grad_h;         /* out [rows][D] or NULL for the loss only: d(scale * lik)/dh */
  float* grad_w1;        /* acc, with grad_h */
  float* grad_b1;        /* acc */
  float* grad_w2;        /* acc */
  float* grad_b2;        /* acc */
  void* workspace;       /* >= hode_readout_mlp_workspace_bytes */
  size_t workspace_bytes;
} hode_readout_mlp_desc;

/* Ensemble CRPS of posterior samples (reference training_utils.py:147-176 / :247-264: mc_itr decoder passes stacked to
 * (T', B, obs, M) and properscoring.crps_ensemble per element).  Member m of (time t, patient b) is the latent_dim-vector
 * at h + t * time_stride + m * member_stride + b * patient_stride (floats); with w != NULL the scored value of component
 * o is w[o] . vector + b[o] (the decoder's linear readout, never materialised), with w == NULL it is vector[o]. */
typedef struct hode_crps_desc {
  uint32_t struct_size;
  int32_t n_times;        /* T' scored time points */
  int32_t batch;          /* B patients */
  int32_t n_members;      /* M ensemble members, 1..128 */
  int32_t latent_dim;     /* width of a member vector, <= 128 */
  int32_t obs_dim;        /* scored components per (time, patient), <= 128 */
  int64_t time_stride;
  int64_t member_stride;
  int64_t patient_stride;
  const float* h;
  const float* w;         /* [obs][latent_dim] or NULL */
  const float* b;         /* [obs] or NULL */
  const float* truth;     /* [T'][B][obs] */
  float* crps;            /* out [T'][B][obs], or NULL */
  float* crps_sum;        /* out [T'][B]: sum over the obs components, or NULL */
} hode_crps_desc;

/* Monte-Carlo KL of the diagonal-Gaussian posterior against Exponential(rate) (reference model.py:1198-1214 with
 * :18-31 and :41-45): kl[i] = mean_s ( log N(z_s; mu, sigma) - log Exp(z_s; rate) ) for element i = (patient, latent
 * component), z_s = noise[s][i] * sigma + mu with non-positive draws replaced by clamp_value (no gradient through them). */
typedef struct hode_mckl_desc {
  uint32_t struct_size;
  int32_t n_samples;     /* S */
  int64_t rows;          /* B * D elements */
  float rate;            /* 100 in the reference */
  float clamp_value;     /* torch.finfo(float32).eps in the reference */
  const float* mu;       /* [rows] */
  const float* log_var;  /* [rows] */
  const float* noise;    /* [S][rows] standard-normal draws */
  float* kl;             /* out [rows] */
  float* grad_mu;        /* out [rows] d kl[i] / d mu[i], or NULL */
  float* grad_log_var;   /* out [rows] d kl[i] / d log_var[i], or NULL */
} hode_mckl_desc;

#define HODE_WS_RK_FWD 0
#define HODE_WS_RK_BWD 1
#define HODE_WS_DOPRI5_FWD 2
#define HODE_WS_DOPRI5_BWD 3

int hode_version(void);
const char* hode_last_error_string(void);

/* HODE_RHS_NEURAL backward.  With grad_w1 / grad_b1 / grad_w2 / grad_b2 set (accumulators, [10D][D+1] / [10D] / [D][10D] /
 * [D]) hode_rk_bwd accumulates the weight gradients itself -- outer products over patients on the matrix cores, one
 * partial block per wave folded in a fixed order -- and the workspace is a few MB.  With grad_w1 == NULL it fills grad_y0
 * only and writes, per (step, stage) instance, the four operands of the weight-gradient GEMMs patient-minor into the
 * workspace: a1t[inst][10D][B], u1t[inst][10D][B], yet[inst][D+1][B], u2t[inst][D][B] (inst = (T-1) * stages); this returns
 * their byte offsets and the caller contracts them: grad_w1 = sum_inst u1t yet^T, grad_b1 = sum u1t, grad_w2 = sum_inst u2t
 * a1t^T, grad_b2 = sum u2t (the one-patient-per-lane kernels, HODE_NEURAL_LAYOUT=t, always work this way). */
int hode_neural_tape_offsets(const hode_solve_desc* desc, size_t* out4);

/* HODE_RHS_ROCHE_REAL backward: hode_rk_bwd fills grad_y0 and grad_theta[0..2] (k_immunity, kel, kel2) and tapes the
 * weight-gradient GEMM operands at the start of the workspace as tape[inst][rows][B], inst = (T-1)*stages,
 * rows = 5 + 4H + 5(D-4) in the order Y3(3) A11(H) U11(H) U12(1) A21(H) U21(H) U22(1) HH RH UR UZ UH (D-4 each). */

/* bytes of device scratch the given entry point needs for this descriptor (0 if none) */
size_t hode_workspace_bytes(const hode_solve_desc* desc, int which);
size_t hode_lstm_workspace_bytes(const hode_lstm_desc* desc);

/* fixed-grid solve: h[0] = y0, h[n+1] = h[n] + step(h[n]); one launch runs the whole time loop */
int hode_rk_fwd(const hode_solve_desc* desc, void* hip_stream);
/* discrete adjoint of hode_rk_fwd: reverse sweep that recomputes the stages of every step from h[n] */
int hode_rk_bwd(const hode_solve_desc* desc, void* hip_stream);

/* adaptive Dormand-Prince 5(4), torchdiffeq 0.2.2 semantics (batch-global controller, dense output).
 * Synchronises the stream once per chunk of attempts to read the controller record.
 * rhs kinds: HODE_RHS_ROCHE / _ABLATE (latent_dim 4, 6, 8, 12) and HODE_RHS_NEURAL (latent_dim 6, 8, 12, hidden_dim = 10 D;
 * the backward accumulates into grad_w1 [10D][D+1], grad_b1 [10D], grad_w2 [D][10D], grad_b2 [D] -- no operand tape). */
int hode_dopri5_fwd(const hode_solve_desc* desc, void* hip_stream);
int hode_dopri5_bwd(const hode_solve_desc* desc, void* hip_stream);

/* What hode_dopri5_fwd leaves in its workspace for the backward and for inspection.  out5 receives the byte offsets of:
 * [0] a hode_dopri5_init_record, [1] tape_t double[n_accepted] (start time of accepted step n), [2] tape_dt
 * double[n_accepted], [3] tape_j int32[2 n_accepted] (first / one-past-last output index interpolated inside step n),
 * [4] tape_y float[n_accepted + 1][B][D] (state at the start of step n). */
typedef struct hode_dopri5_init_record {
  float h0, d0, d1, d2, h1;  /* Hairer's initial step selection: dt_0 = min(100 h0, h1) (torchdiffeq _select_initial_step) */
  int32_t first_accepted;    /* 1: the attempt that ran with dt_0 was accepted, i.e. dt_0 is on the tape and carries gradient */
  float sigma;               /* written by hode_dopri5_bwd: d loss / d dt_0 */
  int32_t pad;
} hode_dopri5_init_record;
int hode_dopri5_tape_offsets(const hode_solve_desc* desc, size_t* out5);

size_t hode_readout_workspace_bytes(const hode_readout_desc* desc);
int hode_readout_sse(const hode_readout_desc* desc, void* hip_stream);

size_t hode_readout_mlp_workspace_bytes(const hode_readout_mlp_desc* desc);
int hode_readout_mlp_sse(const hode_readout_mlp_desc* desc, void* hip_stream);

/* CRPS = 1/M sum_m |x_m - y| - 1/M^2 sum_{i<j} |x_i - x_j| per scored element; one workgroup per (time, patient) */
int hode_ensemble_crps(const hode_crps_desc* desc, void* hip_stream);

int hode_mc_kl_exponential(const hode_mckl_desc* desc, void* hip_stream);

int hode_lstm_fwd(const hode_lstm_desc* desc, void* hip_stream);
int hode_lstm_bwd(const hode_lstm_desc* desc, void* hip_stream);
/* h_prev[t][b][0 .. obs_dim) = x * mask (x when mask is NULL): the columns of the weight-gradient operand rows that
 * hode_lstm_bwd leaves to the caller.  Independent of the recurrence: may run on another stream beside hode_lstm_bwd, before
 * the product that reads h_prev.  Takes the descriptor of the hode_lstm_bwd call it belongs to (validated as a whole); reads
 * seq_len, batch, input_dim, hidden_dim, obs_dim, x, mask and writes h_prev. */
int hode_lstm_fill_operand(const hode_lstm_desc* desc, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* HODE_H_ */
