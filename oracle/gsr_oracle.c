/*
 * gsr_oracle.c -- ORACLE O2: fp32, tile-based CPU restatement of the differentiable
 * 3D-Gaussian rasterizer (RGB + depth + radii, forward and backward).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (gs-dynamics_amd/) never
 * links, imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED: the algorithm it restates lives in the third-party CUDA extension
 * JonathonLuiten/diff-gaussian-rasterization-w-depth (a fork of
 * graphdeco-inria/diff-gaussian-rasterization), which gs-dynamics installs from an unpinned
 * git HEAD (/root/reference/README.md:28-32) and which is NOT present under /root/reference.
 * The reference ships no tests or golden vectors for this path.  What is restated here is the
 * published algorithm (3DGS paper sec. 4-6 + EWA splatting) with the conventions listed in
 * SURVEY.md Appendix A (A-1..A-9); it is anchored on the reference's call sites:
 *   - settings record built at  /root/reference/src/tracking/helpers.py:10-33
 *   - rasterizer invoked at     /root/reference/src/tracking/train_utils.py:174-192
 *   - forward-only use at       /root/reference/src/render/renderer.py:18-23
 *   - consumers of radii/means2D.grad at /root/reference/src/tracking/external.py:138-142
 * It is cross-checked against ORACLE O1 (oracle/dense_oracle.py: fp64, dense, PyTorch
 * autograd -- no hand-derived backward) and against closed-form known-answer cases in tests/.
 *
 * Every float op here is a plain IEEE fp32 mul/add/div/sqrt in a fixed order (compile with
 * -ffp-contract=off); the HIP preprocess kernel uses the same order with contraction off, so
 * integer outputs (radii, tile rects, sort keys, sorted lists) can be compared bit-exactly.
 *
 * Pipeline (names follow the domain: Gaussians, tiles, duplicates/entries, ranges):
 *   preprocess -> per-Gaussian offsets -> (tile,depth) keyed entries -> stable sort ->
 *   per-tile ranges -> per-tile front-to-back blend            (forward)
 *   per-tile back-to-front replay -> per-entry partial grads -> per-Gaussian reduce ->
 *   conic/cov2D/cov3D/projection chain rule                     (backward)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float gsro_f32;   /* a genuine binary32, whatever `float` means below */
#ifdef GSRO_F64
/* fp64 BUILD of the same restatement (oracle/Makefile: libgsr_oracle64.so; tests only): every `float` below is a double and every
 * libm call its double form, so the identical sequence of operations is evaluated with 53-bit significands.  Used to tell how much of
 * a difference between two fp32 evaluations (the HIP kernels and the fp32 build of this file) is conditioning: a Gaussian whose
 * gradient is a small difference of large terms is off in ANY fp32 order.  The discrete decisions -- radii, tile rects, depth sort
 * keys -- can be taken over from an fp32 run (gsro_set_overrides), so that both builds blend the same lists. */
#define float double
#define sqrtf sqrt
#define expf exp
#define logf log
#define ceilf ceil
#define floorf floor
#define fabsf fabs
#define fmaxf fmax
#define fminf fmin
#define powf pow
#endif
/* optional per-Gaussian decisions of another (fp32) run: radii [P], rect [P,4], binary32 depths [P] (NULL: computed here) */
static const int32_t *g_over_radii = 0, *g_over_rect = 0;
static const gsro_f32 *g_over_depth32 = 0;
void gsro_set_overrides(const int32_t *radii, const int32_t *rect, const gsro_f32 *depth32) {
  g_over_radii = radii; g_over_rect = rect; g_over_depth32 = depth32;
}

#define TILE 16
#define NEAR_Z 0.2f
#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f
#define T_EPS 0.0001f

typedef struct {
  int H, W;
  float tanfovx, tanfovy;
  float scale_modifier;
  int sh_degree;     /* active degree */
  int M;             /* coefficients per Gaussian in shs ((max_degree+1)^2), 0 if unused */
  int prefiltered;   /* accepted, ignored like the reference call sites (always False) */
  float bg[3];
  float view[16];    /* the 16 floats of settings.viewmatrix (= w2c^T row-major, i.e. column-major w2c) */
  float proj[16];    /* the 16 floats of settings.projmatrix */
  float campos[3];
} gsro_camera;

typedef struct {
  gsro_camera cam;
  int P, T, gx, gy;
  uint32_t D;
  /* inputs (borrowed pointers, must outlive the ctx) */
  const float *means3D, *scales, *rot, *opac, *colors_precomp, *shs, *cov3D_precomp;
  /* geometry state */
  float *means2D;        /* [P,2] pixel means */
  float *depth;          /* [P] view z */
  float *conic_opacity;  /* [P,4] A,B,C,opacity */
  float *rgb;            /* [P,3] colour fed to the blend */
  uint8_t *clamped;      /* [P,3] SH clamp flags */
  float *cov3D;          /* [P,6] */
  int32_t *radii;        /* [P] */
  int32_t *rect;         /* [P,4] minx,miny,maxx,maxy in tiles */
  uint32_t *tiles_touched;
  uint32_t *offsets;     /* [P+1] exclusive prefix of tiles_touched */
  /* binning state */
  uint64_t *keys;        /* [D] sorted keys */
  uint32_t *point_list;  /* [D] sorted Gaussian indices */
  uint32_t *ranges;      /* [T,2] */
  /* image state */
  float *final_T;        /* [H*W] */
  uint32_t *n_contrib;   /* [H*W] */
  uint8_t *ambiguous;    /* [H*W] 1 if any threshold decision was within rel. 1e-5 (+ the rounding slack of a cancelling exponent) of flipping */
  float *out_color;      /* [3,H,W] */
  float *out_depth;      /* [H*W] */
} gsro_ctx;

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int f2i_sat(float v) {  /* C (int) cast with the saturation a GPU cvt does */
  if (!(v == v)) return 0;
  if (v >= 2147483520.0f) return 2147483647;
  if (v <= -2147483648.0f) return (int)(-2147483647 - 1);
  return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---- SH basis constants (real spherical harmonics, degrees 0..3) ---- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* Appendix A.1-9: SH -> RGB along normalize(mean - campos), +0.5, clamp at 0 with flags. */
static void sh_to_rgb(const gsro_ctx *c, int i, float out[3], uint8_t clamped[3]) {
  const float *p = c->means3D + 3 * i;
  float dx = p[0] - c->cam.campos[0], dy = p[1] - c->cam.campos[1], dz = p[2] - c->cam.campos[2];
  float len = sqrtf(dx * dx + dy * dy + dz * dz);
  float inv = 1.0f / len;
  float x = dx * inv, y = dy * inv, z = dz * inv;
  const float *sh = c->shs + (size_t)i * c->cam.M * 3;
  int deg = c->cam.sh_degree;
  for (int ch = 0; ch < 3; ++ch) {
#define S(k) sh[(k)*3 + ch]
    float r = SH_C0 * S(0);
    if (deg > 0) {
      r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
            SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
        if (deg > 2) {
          r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
              SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
              SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
        }
      }
    }
#undef S
    r += 0.5f;
    clamped[ch] = (r < 0.0f);
    out[ch] = r < 0.0f ? 0.0f : r;
  }
}

/* R(q) S^2 R(q)^T, six upper-triangular entries (Appendix A.1-3). q = (r,x,y,z), not renormalised. */
static void cov3d_from_scale_rot(const float s_in[3], float mod, const float q[4], float cov[6]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                   {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                   {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  float s[3] = {mod * s_in[0], mod * s_in[1], mod * s_in[2]};
  float M[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * s[j];
  /* Sigma = M M^T */
  float S00 = M[0][0] * M[0][0] + M[0][1] * M[0][1] + M[0][2] * M[0][2];
  float S01 = M[0][0] * M[1][0] + M[0][1] * M[1][1] + M[0][2] * M[1][2];
  float S02 = M[0][0] * M[2][0] + M[0][1] * M[2][1] + M[0][2] * M[2][2];
  float S11 = M[1][0] * M[1][0] + M[1][1] * M[1][1] + M[1][2] * M[1][2];
  float S12 = M[1][0] * M[2][0] + M[1][1] * M[2][1] + M[1][2] * M[2][2];
  float S22 = M[2][0] * M[2][0] + M[2][1] * M[2][1] + M[2][2] * M[2][2];
  cov[0] = S00; cov[1] = S01; cov[2] = S02; cov[3] = S11; cov[4] = S12; cov[5] = S22;
}

/* The 2x3 matrix T = J * W_rot of the EWA projection, with the frustum clamp of A.1-4.
 * Returns clamped tx,ty and the clamp-active flags (used by backward, convention A-3). */
typedef struct { float T[2][3]; float tx, ty, tz; int xclamped, yclamped; float fx, fy; } ewa_t;
static void ewa_setup(const gsro_camera *cam, const float pv[3], ewa_t *e) {
  float fx = (float)cam->W / (2.0f * cam->tanfovx);
  float fy = (float)cam->H / (2.0f * cam->tanfovy);
  float limx = 1.3f * cam->tanfovx, limy = 1.3f * cam->tanfovy;
  float tz = pv[2];
  float txtz = pv[0] / tz, tytz = pv[1] / tz;
  e->xclamped = (txtz < -limx) || (txtz > limx);
  e->yclamped = (tytz < -limy) || (tytz > limy);
  float tx = clampf(txtz, -limx, limx) * tz;
  float ty = clampf(tytz, -limy, limy) * tz;
  float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  const float *V = cam->view; /* W_rot(r,c) = V[c*4+r] */
  for (int j = 0; j < 3; ++j) {
    float W0j = V[j * 4 + 0], W1j = V[j * 4 + 1], W2j = V[j * 4 + 2];
    e->T[0][j] = J00 * W0j + J02 * W2j;
    e->T[1][j] = J11 * W1j + J12 * W2j;
  }
  e->tx = tx; e->ty = ty; e->tz = tz; e->fx = fx; e->fy = fy;
}

static void preprocess_one(gsro_ctx *c, int i) {
  const gsro_camera *cam = &c->cam;
  c->radii[i] = 0;
  c->tiles_touched[i] = 0;
  c->rect[4 * i + 0] = c->rect[4 * i + 1] = c->rect[4 * i + 2] = c->rect[4 * i + 3] = 0;
  const float *p = c->means3D + 3 * i;
  const float *V = cam->view, *Pm = cam->proj;
  float pv[3];
  pv[0] = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
  pv[1] = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
  pv[2] = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  if (pv[2] <= NEAR_Z) return; /* A.1-1 */
  float hx = Pm[0] * p[0] + Pm[4] * p[1] + Pm[8] * p[2] + Pm[12];
  float hy = Pm[1] * p[0] + Pm[5] * p[1] + Pm[9] * p[2] + Pm[13];
  float hw = Pm[3] * p[0] + Pm[7] * p[1] + Pm[11] * p[2] + Pm[15];
  float pw = 1.0f / (hw + 0.0000001f); /* A.1-2 */
  float ndcx = hx * pw, ndcy = hy * pw;

  float cov6[6];
  if (c->cov3D_precomp) memcpy(cov6, c->cov3D_precomp + 6 * i, sizeof cov6);
  else cov3d_from_scale_rot(c->scales + 3 * i, cam->scale_modifier, c->rot + 4 * i, cov6);
  memcpy(c->cov3D + 6 * i, cov6, sizeof cov6);

  ewa_t e;
  ewa_setup(cam, pv, &e);
  /* cov2D = T Sigma T^T (A.1-4) */
  float S[3][3] = {{cov6[0], cov6[1], cov6[2]}, {cov6[1], cov6[3], cov6[4]}, {cov6[2], cov6[4], cov6[5]}};
  float U[2][3]; /* U = T Sigma */
  for (int r = 0; r < 2; ++r)
    for (int k = 0; k < 3; ++k) U[r][k] = e.T[r][0] * S[0][k] + e.T[r][1] * S[1][k] + e.T[r][2] * S[2][k];
  float a = U[0][0] * e.T[0][0] + U[0][1] * e.T[0][1] + U[0][2] * e.T[0][2];
  float b = U[0][0] * e.T[1][0] + U[0][1] * e.T[1][1] + U[0][2] * e.T[1][2];
  float cc = U[1][0] * e.T[1][0] + U[1][1] * e.T[1][1] + U[1][2] * e.T[1][2];
  a += 0.3f; cc += 0.3f;
  float det = a * cc - b * b;
  if (det == 0.0f) return; /* A.1-5 */
  float det_inv = 1.0f / det;
  float cA = cc * det_inv, cB = -b * det_inv, cC = a * det_inv;
  float mid = 0.5f * (a + cc);
  float disc = mid * mid - det;
  float sq = sqrtf(disc > 0.1f ? disc : 0.1f);
  float l1 = mid + sq, l2 = mid - sq;
  float radius = ceilf(3.0f * sqrtf(l1 > l2 ? l1 : l2)); /* A.1-6 */
  float px = ((ndcx + 1.0f) * (float)cam->W - 1.0f) * 0.5f; /* A.1-7 */
  float py = ((ndcy + 1.0f) * (float)cam->H - 1.0f) * 0.5f;
  /* A.1-8 tile rect, C int truncation */
  int minx = imin(c->gx, imax(0, f2i_sat((px - radius) / (float)TILE)));
  int miny = imin(c->gy, imax(0, f2i_sat((py - radius) / (float)TILE)));
  int maxx = imin(c->gx, imax(0, f2i_sat((px + radius + (float)(TILE - 1)) / (float)TILE)));
  int maxy = imin(c->gy, imax(0, f2i_sat((py + radius + (float)(TILE - 1)) / (float)TILE)));
  if (g_over_rect) {   /* the other run's decisions: same Gaussians visible, same rects, same radii */
    minx = g_over_rect[4 * i + 0]; miny = g_over_rect[4 * i + 1]; maxx = g_over_rect[4 * i + 2]; maxy = g_over_rect[4 * i + 3];
    radius = (float)g_over_radii[i];
  }
  if ((maxx - minx) * (maxy - miny) == 0) return;

  if (c->colors_precomp) {
    for (int ch = 0; ch < 3; ++ch) { c->rgb[3 * i + ch] = c->colors_precomp[3 * i + ch]; c->clamped[3 * i + ch] = 0; }
  } else {
    sh_to_rgb(c, i, c->rgb + 3 * i, c->clamped + 3 * i);
  }
  c->depth[i] = pv[2];
  c->radii[i] = f2i_sat(radius);
  c->means2D[2 * i] = px; c->means2D[2 * i + 1] = py;
  c->conic_opacity[4 * i + 0] = cA; c->conic_opacity[4 * i + 1] = cB;
  c->conic_opacity[4 * i + 2] = cC; c->conic_opacity[4 * i + 3] = c->opac[i];
  c->rect[4 * i + 0] = minx; c->rect[4 * i + 1] = miny; c->rect[4 * i + 2] = maxx; c->rect[4 * i + 3] = maxy;
  c->tiles_touched[i] = (uint32_t)((maxx - minx) * (maxy - miny));
}

/* stable LSD radix sort of (key,value) pairs on the low `bits` bits (A.2) */
static void radix_sort_pairs(uint64_t *k, uint32_t *v, uint32_t n, int bits) {
  uint64_t *k2 = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
  uint32_t *v2 = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (int shift = 0; shift < bits; shift += 8) {
    uint32_t cnt[257];
    memset(cnt, 0, sizeof cnt);
    for (uint32_t i = 0; i < n; ++i) cnt[((k[i] >> shift) & 0xff) + 1]++;
    for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t d = (uint32_t)((k[i] >> shift) & 0xff);
      k2[cnt[d]] = k[i]; v2[cnt[d]] = v[i]; cnt[d]++;
    }
    uint64_t *tk = k; k = k2; k2 = tk;
    uint32_t *tv = v; v = v2; v2 = tv;
  }
  /* even number of swaps needed to land in the caller's buffers */
  int passes = (bits + 7) / 8;
  if (passes & 1) { memcpy(k2, k, sizeof(uint64_t) * n); memcpy(v2, v, sizeof(uint32_t) * n); free(k); free(v); }
  else { free(k2); free(v2); }
}

static int near_rel(float v, float thr) { return fabsf(v - thr) <= 1e-5f * fabsf(thr); }

static void render_tile_fwd(gsro_ctx *c, int tile) {
  const gsro_camera *cam = &c->cam;
  int tx0 = (tile % c->gx) * TILE, ty0 = (tile / c->gx) * TILE;
  uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
  for (int ly = 0; ly < TILE; ++ly)
    for (int lx = 0; lx < TILE; ++lx) {
      int px = tx0 + lx, py = ty0 + ly;
      if (px >= cam->W || py >= cam->H) continue;
      int pix = py * cam->W + px;
      float pxf = (float)px, pyf = (float)py; /* A-8: pixel centres at integers */
      float T = 1.0f, C[3] = {0, 0, 0}, Dp = 0.0f;
      uint32_t contributor = 0, last = 0;
      uint8_t amb = 0;
      for (uint32_t s = r0; s < r1; ++s) {
        contributor++;
        uint32_t g = c->point_list[s];
        float dx = c->means2D[2 * g] - pxf, dy = c->means2D[2 * g + 1] - pyf;
        const float *co = c->conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        /* What ANY fp32 evaluation order can move the exponent by: its three terms cancel (a nearly singular conic far from its
           centre: terms of +-400 that sum to -5), so the decisions taken on it -- power > 0, alpha >= 1/255 -- are ambiguous within
           a few roundings of the terms' magnitudes, not within a fixed 1e-5 (soak seed 4242 case 126: alpha 1.3e-5 above 1/255 in
           fp64, 5.5e-6 below it with the conic pre-scaled by log2 e; terms' magnitudes 801). */
        float slack = 4.0f * 5.9604645e-8f * (0.5f * fabsf(co[0]) * dx * dx + 0.5f * fabsf(co[2]) * dy * dy + fabsf(co[1] * dx * dy));
        if (fabsf(power) <= slack && co[3] >= ALPHA_MIN) amb = 1;
        if (power > 0.0f) continue;
        float alpha = co[3] * expf(power);
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        if (fabsf(alpha - ALPHA_MIN) <= (1e-5f + slack) * ALPHA_MIN) amb = 1;
        if (alpha < ALPHA_MIN) continue;
        float test_T = T * (1.0f - alpha);
        if (near_rel(test_T, T_EPS)) amb = 1;
        if (test_T < T_EPS) break; /* this Gaussian is NOT blended (A.3) */
        float w = alpha * T;
        for (int ch = 0; ch < 3; ++ch) C[ch] += c->rgb[3 * g + ch] * w;
        Dp += c->depth[g] * w;
        T = test_T;
        last = contributor;
      }
      c->final_T[pix] = T;
      c->n_contrib[pix] = last;
      c->ambiguous[pix] = amb;
      for (int ch = 0; ch < 3; ++ch) c->out_color[ch * cam->H * cam->W + pix] = C[ch] + T * cam->bg[ch];
      c->out_depth[pix] = Dp; /* no background term, not normalised (A.3, [M]) */
    }
}

void gsro_free(gsro_ctx *c) {
  if (!c) return;
  free(c->means2D); free(c->depth); free(c->conic_opacity); free(c->rgb); free(c->clamped);
  free(c->cov3D); free(c->radii); free(c->rect); free(c->tiles_touched); free(c->offsets);
  free(c->keys); free(c->point_list); free(c->ranges); free(c->final_T); free(c->n_contrib);
  free(c->ambiguous); free(c->out_color); free(c->out_depth);
  free(c);
}

static int ceil_log2_u32(uint32_t n) { int b = 0; while (((uint64_t)1 << b) < n) ++b; return b; }

gsro_ctx *gsro_forward(const gsro_camera *cam, int P, const float *means3D, const float *scales,
                       const float *rot, const float *opac, const float *colors_precomp,
                       const float *shs, const float *cov3D_precomp, int nthreads) {
  gsro_ctx *c = (gsro_ctx *)calloc(1, sizeof(gsro_ctx));
  c->cam = *cam; c->P = P;
  c->gx = (cam->W + TILE - 1) / TILE; c->gy = (cam->H + TILE - 1) / TILE; c->T = c->gx * c->gy;
  c->means3D = means3D; c->scales = scales; c->rot = rot; c->opac = opac;
  c->colors_precomp = colors_precomp; c->shs = shs; c->cov3D_precomp = cov3D_precomp;
  size_t Pn = P > 0 ? (size_t)P : 1, N = (size_t)cam->H * cam->W;
  c->means2D = (float *)calloc(2 * Pn, sizeof(float)); c->depth = (float *)calloc(Pn, sizeof(float));
  c->conic_opacity = (float *)calloc(4 * Pn, sizeof(float)); c->rgb = (float *)calloc(3 * Pn, sizeof(float));
  c->clamped = (uint8_t *)calloc(3 * Pn, 1); c->cov3D = (float *)calloc(6 * Pn, sizeof(float));
  c->radii = (int32_t *)calloc(Pn, 4); c->rect = (int32_t *)calloc(4 * Pn, 4);
  c->tiles_touched = (uint32_t *)calloc(Pn, 4); c->offsets = (uint32_t *)calloc(Pn + 1, 4);
  c->ranges = (uint32_t *)calloc(2 * (size_t)c->T, 4);
  c->final_T = (float *)calloc(N, sizeof(float)); c->n_contrib = (uint32_t *)calloc(N, 4);
  c->ambiguous = (uint8_t *)calloc(N, 1);
  c->out_color = (float *)calloc(3 * N, sizeof(float)); c->out_depth = (float *)calloc(N, sizeof(float));
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; ++i) preprocess_one(c, i);
  uint32_t acc = 0;
  for (int i = 0; i < P; ++i) { c->offsets[i] = acc; acc += c->tiles_touched[i]; }
  c->offsets[P] = acc; c->D = acc;
  c->keys = (uint64_t *)malloc(sizeof(uint64_t) * (acc ? acc : 1));
  c->point_list = (uint32_t *)malloc(sizeof(uint32_t) * (acc ? acc : 1));
  /* A.2: key = tile << 32 | depth bits, emitted row-major over the rect */
  for (int i = 0; i < P; ++i) {
    if (c->radii[i] <= 0) continue;
    uint32_t off = c->offsets[i];
    gsro_f32 d32 = g_over_depth32 ? g_over_depth32[i] : (gsro_f32)c->depth[i];   /* sort key = binary32 depth bits, in every build */
    uint32_t dbits; memcpy(&dbits, &d32, 4);
    for (int y = c->rect[4 * i + 1]; y < c->rect[4 * i + 3]; ++y)
      for (int x = c->rect[4 * i + 0]; x < c->rect[4 * i + 2]; ++x) {
        uint64_t key = (uint64_t)(uint32_t)(y * c->gx + x);
        key = (key << 32) | dbits;
        c->keys[off] = key; c->point_list[off] = (uint32_t)i; off++;
      }
  }
  radix_sort_pairs(c->keys, c->point_list, c->D, 32 + ceil_log2_u32((uint32_t)c->T));
  for (uint32_t s = 0; s < c->D; ++s) {
    uint32_t t = (uint32_t)(c->keys[s] >> 32);
    if (s == 0) c->ranges[2 * t] = 0;
    else {
      uint32_t pt = (uint32_t)(c->keys[s - 1] >> 32);
      if (pt != t) { c->ranges[2 * pt + 1] = s; c->ranges[2 * t] = s; }
    }
    if (s == c->D - 1) c->ranges[2 * t + 1] = c->D;
  }
  /* background for pixels whose tile list is empty is handled by the same loop */
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < c->T; ++t) render_tile_fwd(c, t);
  return c;
}

/* ------------------------------------------------------------------ backward */

/* Per-tile back-to-front replay (A.4).  Accumulates, per entry of the tile list, the nine
 * partials {dL/dmean2D_pix.xy, dL/dconic.ABC (true partials), dL/dopacity, dL/dcolour.rgb}
 * into part[9*(s)], s = sorted position; pixel order is row-major inside the tile. */
static void render_tile_bwd(const gsro_ctx *c, int tile, const float *dL_dcolor, float *part) {
  const gsro_camera *cam = &c->cam;
  int tx0 = (tile % c->gx) * TILE, ty0 = (tile / c->gx) * TILE;
  uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
  size_t N = (size_t)cam->H * cam->W;
  for (int ly = 0; ly < TILE; ++ly)
    for (int lx = 0; lx < TILE; ++lx) {
      int px = tx0 + lx, py = ty0 + ly;
      if (px >= cam->W || py >= cam->H) continue;
      int pix = py * cam->W + px;
      float pxf = (float)px, pyf = (float)py;
      float T_final = c->final_T[pix];
      float T = T_final;
      uint32_t last = c->n_contrib[pix];
      float dLp[3] = {dL_dcolor[pix], dL_dcolor[N + pix], dL_dcolor[2 * N + pix]};
      float bg_dot = cam->bg[0] * dLp[0] + cam->bg[1] * dLp[1] + cam->bg[2] * dLp[2];
      float accum[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0.0f;
      for (uint32_t k = last; k-- > 0;) { /* entries r0+last-1 .. r0 */
        uint32_t s = r0 + k;
        if (s >= r1) continue;
        uint32_t g = c->point_list[s];
        float dx = c->means2D[2 * g] - pxf, dy = c->means2D[2 * g + 1] - pyf;
        const float *co = c->conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float G = expf(power);
        float alpha = co[3] * G;
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        if (alpha < ALPHA_MIN) continue; /* A-2 */
        T = T / (1.0f - alpha);
        float w = alpha * T;
        float dL_dalpha = 0.0f;
        float *pp = part + 9 * (size_t)s;
        for (int ch = 0; ch < 3; ++ch) {
          float col = c->rgb[3 * g + ch];
          accum[ch] = last_alpha * last_color[ch] + (1.0f - last_alpha) * accum[ch];
          last_color[ch] = col;
          dL_dalpha += (col - accum[ch]) * dLp[ch];
          pp[6 + ch] += w * dLp[ch];
        }
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
        /* A-1: min(0.99,.) is straight-through */
        float dL_dG = co[3] * dL_dalpha;
        float gdx = G * dx, gdy = G * dy;
        float dG_ddx = -gdx * co[0] - gdy * co[1];
        float dG_ddy = -gdy * co[2] - gdx * co[1];
        pp[0] += dL_dG * dG_ddx;           /* d/d mean2D.x (pixel units) */
        pp[1] += dL_dG * dG_ddy;
        pp[2] += -0.5f * gdx * dx * dL_dG; /* dL/dA */
        pp[3] += -gdx * dy * dL_dG;        /* dL/dB (true partial) */
        pp[4] += -0.5f * gdy * dy * dL_dG; /* dL/dC */
        pp[5] += G * dL_dalpha;            /* dL/dopacity */
      }
    }
}

/* Backward.  Outputs (all zero-filled by this function, sizes in floats):
 *   dL_dmeans3D[3P] dL_dmeans2D[3P] (x,y NDC-scaled per A-6, z=0) dL_dcolors[3P] dL_dopacity[P]
 *   dL_dscales[3P] dL_drot[4P] dL_dcov3D[6P] dL_dsh[3*M*P or NULL]                                    */
void gsro_backward(const gsro_ctx *c, const float *dL_dcolor, float *dL_dmeans3D, float *dL_dmeans2D,
                   float *dL_dcolors, float *dL_dopacity, float *dL_dscales, float *dL_drot,
                   float *dL_dcov3D, float *dL_dsh, int nthreads) {
  const gsro_camera *cam = &c->cam;
  int P = c->P;
  memset(dL_dmeans3D, 0, sizeof(float) * 3 * P); memset(dL_dmeans2D, 0, sizeof(float) * 3 * P);
  memset(dL_dcolors, 0, sizeof(float) * 3 * P); memset(dL_dopacity, 0, sizeof(float) * P);
  memset(dL_dscales, 0, sizeof(float) * 3 * P); memset(dL_drot, 0, sizeof(float) * 4 * P);
  memset(dL_dcov3D, 0, sizeof(float) * 6 * P);
  if (dL_dsh) memset(dL_dsh, 0, sizeof(float) * 3 * (size_t)cam->M * P);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  float *part = (float *)calloc(9 * (size_t)(c->D ? c->D : 1), sizeof(float));
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < c->T; ++t) render_tile_bwd(c, t, dL_dcolor, part);
  /* scatter sorted-position partials to Gaussian-major entry slots e = offsets[g] + k,
   * k = row-major rank of the tile inside the Gaussian's rect; then reduce per Gaussian in k order */
  float *acc = (float *)calloc(9 * (size_t)(P ? P : 1), sizeof(float));
  {
    float *gm = (float *)calloc(9 * (size_t)(c->D ? c->D : 1), sizeof(float));
    for (int t = 0; t < c->T; ++t) {
      int tx = t % c->gx, ty = t / c->gx;
      for (uint32_t s = c->ranges[2 * t]; s < c->ranges[2 * t + 1]; ++s) {
        uint32_t g = c->point_list[s];
        const int32_t *r = c->rect + 4 * g;
        uint32_t k = (uint32_t)((ty - r[1]) * (r[2] - r[0]) + (tx - r[0]));
        memcpy(gm + 9 * (size_t)(c->offsets[g] + k), part + 9 * (size_t)s, 9 * sizeof(float));
      }
    }
    for (int g = 0; g < P; ++g)
      for (uint32_t e = c->offsets[g]; e < c->offsets[g + 1]; ++e)
        for (int q = 0; q < 9; ++q) acc[9 * g + q] += gm[9 * (size_t)e + q];
    free(gm);
  }
  free(part);

  const float *V = cam->view, *Pm = cam->proj;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; ++i) {
    if (c->radii[i] <= 0) continue;
    const float *a9 = acc + 9 * i;
    float gmx = a9[0], gmy = a9[1], gA = a9[2], gB = a9[3], gC = a9[4], gop = a9[5];
    const float *p = c->means3D + 3 * i;
    /* ---- colour ---- */
    float dL_drgb[3] = {a9[6], a9[7], a9[8]};
    float dmean_sh[3] = {0, 0, 0};
    if (c->colors_precomp) {
      for (int ch = 0; ch < 3; ++ch) dL_dcolors[3 * i + ch] = dL_drgb[ch];
    } else {
      /* SH backward: clamp mask, coefficient grads, and direction -> mean grads */
      for (int ch = 0; ch < 3; ++ch) if (c->clamped[3 * i + ch]) dL_drgb[ch] = 0.0f;
      float ox = p[0] - cam->campos[0], oy = p[1] - cam->campos[1], oz = p[2] - cam->campos[2];
      float len = sqrtf(ox * ox + oy * oy + oz * oz), inv = 1.0f / len;
      float x = ox * inv, y = oy * inv, z = oz * inv;
      int deg = cam->sh_degree, M = cam->M;
      const float *sh = c->shs + (size_t)i * M * 3;
      float *dsh = dL_dsh + (size_t)i * M * 3;
      float dRdx[3] = {0, 0, 0}, dRdy[3] = {0, 0, 0}, dRdz[3] = {0, 0, 0};
      float basis[16]; memset(basis, 0, sizeof basis);
      basis[0] = SH_C0;
      if (deg > 0) {
        basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
        for (int ch = 0; ch < 3; ++ch) {
          dRdx[ch] = -SH_C1 * sh[3 * 3 + ch]; dRdy[ch] = -SH_C1 * sh[1 * 3 + ch]; dRdz[ch] = SH_C1 * sh[2 * 3 + ch];
        }
        if (deg > 1) {
          float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          basis[4] = SH_C2[0] * xy; basis[5] = SH_C2[1] * yz; basis[6] = SH_C2[2] * (2.0f * zz - xx - yy);
          basis[7] = SH_C2[3] * xz; basis[8] = SH_C2[4] * (xx - yy);
          for (int ch = 0; ch < 3; ++ch) {
#define S(k) sh[(k)*3 + ch]
            dRdx[ch] += SH_C2[0] * y * S(4) + SH_C2[2] * 2.0f * -x * S(6) + SH_C2[3] * z * S(7) + SH_C2[4] * 2.0f * x * S(8);
            dRdy[ch] += SH_C2[0] * x * S(4) + SH_C2[1] * z * S(5) + SH_C2[2] * 2.0f * -y * S(6) + SH_C2[4] * 2.0f * -y * S(8);
            dRdz[ch] += SH_C2[1] * y * S(5) + SH_C2[2] * 2.0f * 2.0f * z * S(6) + SH_C2[3] * x * S(7);
#undef S
          }
          if (deg > 2) {
            basis[9] = SH_C3[0] * y * (3.0f * xx - yy); basis[10] = SH_C3[1] * xy * z;
            basis[11] = SH_C3[2] * y * (4.0f * zz - xx - yy); basis[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            basis[13] = SH_C3[4] * x * (4.0f * zz - xx - yy); basis[14] = SH_C3[5] * z * (xx - yy);
            basis[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            for (int ch = 0; ch < 3; ++ch) {
#define S(k) sh[(k)*3 + ch]
              dRdx[ch] += SH_C3[0] * S(9) * 3.0f * 2.0f * xy + SH_C3[1] * S(10) * yz + SH_C3[2] * S(11) * -2.0f * xy +
                          SH_C3[3] * S(12) * -3.0f * 2.0f * xz + SH_C3[4] * S(13) * (-3.0f * xx + 4.0f * zz - yy) +
                          SH_C3[5] * S(14) * 2.0f * xz + SH_C3[6] * S(15) * 3.0f * (xx - yy);
              dRdy[ch] += SH_C3[0] * S(9) * 3.0f * (xx - yy) + SH_C3[1] * S(10) * xz + SH_C3[2] * S(11) * (-3.0f * yy + 4.0f * zz - xx) +
                          SH_C3[3] * S(12) * -3.0f * 2.0f * yz + SH_C3[4] * S(13) * -2.0f * xy +
                          SH_C3[5] * S(14) * -2.0f * yz + SH_C3[6] * S(15) * -3.0f * 2.0f * xy;
              dRdz[ch] += SH_C3[1] * S(10) * xy + SH_C3[2] * S(11) * 4.0f * 2.0f * yz + SH_C3[3] * S(12) * 3.0f * (2.0f * zz - xx - yy) +
                          SH_C3[4] * S(13) * 4.0f * 2.0f * xz + SH_C3[5] * S(14) * (xx - yy);
#undef S
            }
          }
        }
      }
      int ncoef = (deg + 1) * (deg + 1);
      for (int k = 0; k < ncoef; ++k)
        for (int ch = 0; ch < 3; ++ch) dsh[k * 3 + ch] = basis[k] * dL_drgb[ch];
      float ddx = dRdx[0] * dL_drgb[0] + dRdx[1] * dL_drgb[1] + dRdx[2] * dL_drgb[2];
      float ddy = dRdy[0] * dL_drgb[0] + dRdy[1] * dL_drgb[1] + dRdy[2] * dL_drgb[2];
      float ddz = dRdz[0] * dL_drgb[0] + dRdz[1] * dL_drgb[1] + dRdz[2] * dL_drgb[2];
      /* d normalize(o)/d o applied to (ddx,ddy,ddz) */
      float sum2 = ox * ox + oy * oy + oz * oz;
      float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      dmean_sh[0] = ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
      dmean_sh[1] = (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
      dmean_sh[2] = (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
    }
    dL_dopacity[i] = gop;

    /* ---- conic -> cov2D (A.5; 1e-7 added to det^2 in the reciprocal, [M]) ---- */
    float pv[3];
    pv[0] = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
    pv[1] = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
    pv[2] = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
    ewa_t e;
    ewa_setup(cam, pv, &e);
    const float *cv = c->cov3D + 6 * i;
    float S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
    float U0[3], U1[3]; /* Sigma T0, Sigma T1 */
    for (int k = 0; k < 3; ++k) {
      U0[k] = S[k][0] * e.T[0][0] + S[k][1] * e.T[0][1] + S[k][2] * e.T[0][2];
      U1[k] = S[k][0] * e.T[1][0] + S[k][1] * e.T[1][1] + S[k][2] * e.T[1][2];
    }
    float a = U0[0] * e.T[0][0] + U0[1] * e.T[0][1] + U0[2] * e.T[0][2] + 0.3f;
    float b = U0[0] * e.T[1][0] + U0[1] * e.T[1][1] + U0[2] * e.T[1][2];
    float cc = U1[0] * e.T[1][0] + U1[1] * e.T[1][1] + U1[2] * e.T[1][2] + 0.3f;
    float det = a * cc - b * b;
    float d2inv = 1.0f / (det * det + 0.0000001f);
    float dL_da = d2inv * (-cc * cc * gA + b * cc * gB - b * b * gC);
    float dL_dc = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
    float dL_db = d2inv * (2.0f * b * cc * gA - (det + 2.0f * b * b) * gB + 2.0f * a * b * gC);
    /* cov2D -> cov3D (unique-parameter gradients of the symmetric Sigma) */
    float *dcv = dL_dcov3D + 6 * i;
    const float(*T)[3] = e.T;
    dcv[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
    dcv[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
    dcv[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
    dcv[1] = 2.0f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2.0f * T[1][0] * T[1][1] * dL_dc;
    dcv[2] = 2.0f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2.0f * T[1][0] * T[1][2] * dL_dc;
    dcv[4] = 2.0f * T[0][1] * T[0][2] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2.0f * T[1][1] * T[1][2] * dL_dc;
    /* cov2D -> T -> J -> t (view-space mean) */
    float dT0[3], dT1[3];
    for (int j = 0; j < 3; ++j) {
      dT0[j] = 2.0f * U0[j] * dL_da + U1[j] * dL_db;
      dT1[j] = 2.0f * U1[j] * dL_dc + U0[j] * dL_db;
    }
    /* T_rj = sum_c J_rc W_cj  ->  dJ_rc = sum_j dT_rj W_cj ; W_cj = V[j*4+c] */
    float dJ00 = dT0[0] * V[0] + dT0[1] * V[4] + dT0[2] * V[8];
    float dJ02 = dT0[0] * V[2] + dT0[1] * V[6] + dT0[2] * V[10];
    float dJ11 = dT1[0] * V[1] + dT1[1] * V[5] + dT1[2] * V[9];
    float dJ12 = dT1[0] * V[2] + dT1[1] * V[6] + dT1[2] * V[10];
    float tz = 1.0f / e.tz, tz2 = tz * tz, tz3 = tz2 * tz;
    float xm = e.xclamped ? 0.0f : 1.0f, ym = e.yclamped ? 0.0f : 1.0f; /* A-3 */
    float dtx = xm * -e.fx * tz2 * dJ02;
    float dty = ym * -e.fy * tz2 * dJ12;
    float dtz = -e.fx * tz2 * dJ00 - e.fy * tz2 * dJ11 + (2.0f * e.fx * e.tx) * tz3 * dJ02 + (2.0f * e.fy * e.ty) * tz3 * dJ12;
    float dm[3];
    dm[0] = V[0] * dtx + V[1] * dty + V[2] * dtz;
    dm[1] = V[4] * dtx + V[5] * dty + V[6] * dtz;
    dm[2] = V[8] * dtx + V[9] * dty + V[10] * dtz;
    /* ---- 2D mean -> 3D mean through the 4x4 projection (A.5, A-6, A-7) ---- */
    float dL_dm2x = gmx * 0.5f * (float)cam->W, dL_dm2y = gmy * 0.5f * (float)cam->H;
    dL_dmeans2D[3 * i + 0] = dL_dm2x; dL_dmeans2D[3 * i + 1] = dL_dm2y;
    float hx = Pm[0] * p[0] + Pm[4] * p[1] + Pm[8] * p[2] + Pm[12];
    float hy = Pm[1] * p[0] + Pm[5] * p[1] + Pm[9] * p[2] + Pm[13];
    float hw = Pm[3] * p[0] + Pm[7] * p[1] + Pm[11] * p[2] + Pm[15];
    float mw = 1.0f / (hw + 0.0000001f);
    float mul1 = hx * mw * mw, mul2 = hy * mw * mw;
    dm[0] += (Pm[0] * mw - Pm[3] * mul1) * dL_dm2x + (Pm[1] * mw - Pm[3] * mul2) * dL_dm2y;
    dm[1] += (Pm[4] * mw - Pm[7] * mul1) * dL_dm2x + (Pm[5] * mw - Pm[7] * mul2) * dL_dm2y;
    dm[2] += (Pm[8] * mw - Pm[11] * mul1) * dL_dm2x + (Pm[9] * mw - Pm[11] * mul2) * dL_dm2y;
    for (int k = 0; k < 3; ++k) dL_dmeans3D[3 * i + k] = dm[k] + dmean_sh[k];
    /* ---- cov3D -> scale, rotation ---- */
    if (!c->cov3D_precomp) {
      const float *q = c->rot + 4 * i;
      float r = q[0], x = q[1], y = q[2], z = q[3];
      float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                       {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                       {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
      float mod = cam->scale_modifier;
      float s[3] = {mod * c->scales[3 * i], mod * c->scales[3 * i + 1], mod * c->scales[3 * i + 2]};
      /* full symmetric dL/dSigma (off-diagonals halved), dL/dM = 2 dSigma M, M = R diag(s) */
      float dS[3][3] = {{dcv[0], 0.5f * dcv[1], 0.5f * dcv[2]}, {0.5f * dcv[1], dcv[3], 0.5f * dcv[4]}, {0.5f * dcv[2], 0.5f * dcv[4], dcv[5]}};
      float dM[3][3];
      for (int ii = 0; ii < 3; ++ii)
        for (int jj = 0; jj < 3; ++jj)
          dM[ii][jj] = 2.0f * (dS[ii][0] * R[0][jj] * s[jj] + dS[ii][1] * R[1][jj] * s[jj] + dS[ii][2] * R[2][jj] * s[jj]);
      float G[3][3];
      for (int jj = 0; jj < 3; ++jj) {
        dL_dscales[3 * i + jj] = mod * (R[0][jj] * dM[0][jj] + R[1][jj] * dM[1][jj] + R[2][jj] * dM[2][jj]);
        for (int ii = 0; ii < 3; ++ii) G[ii][jj] = dM[ii][jj] * s[jj];
      }
      dL_drot[4 * i + 0] = 2.0f * (-z * G[0][1] + y * G[0][2] + z * G[1][0] - x * G[1][2] - y * G[2][0] + x * G[2][1]);
      dL_drot[4 * i + 1] = 2.0f * (y * G[0][1] + z * G[0][2] + y * G[1][0] - 2.0f * x * G[1][1] - r * G[1][2] + z * G[2][0] + r * G[2][1] - 2.0f * x * G[2][2]);
      dL_drot[4 * i + 2] = 2.0f * (-2.0f * y * G[0][0] + x * G[0][1] + r * G[0][2] + x * G[1][0] + z * G[1][2] - r * G[2][0] + z * G[2][1] - 2.0f * y * G[2][2]);
      dL_drot[4 * i + 3] = 2.0f * (-2.0f * z * G[0][0] - r * G[0][1] + x * G[0][2] + r * G[1][0] - 2.0f * z * G[1][1] + y * G[1][2] + x * G[2][0] + y * G[2][1]);
    }
  }
  free(acc);
}

/* ---- accessors for tests (copy-out) ---- */
uint32_t gsro_num_rendered(const gsro_ctx *c) { return c->D; }
int gsro_num_tiles(const gsro_ctx *c) { return c->T; }
const float *gsro_out_color(const gsro_ctx *c) { return c->out_color; }
const float *gsro_out_depth(const gsro_ctx *c) { return c->out_depth; }
const int32_t *gsro_radii(const gsro_ctx *c) { return c->radii; }
const float *gsro_means2D(const gsro_ctx *c) { return c->means2D; }
const float *gsro_depths(const gsro_ctx *c) { return c->depth; }
const float *gsro_conic_opacity(const gsro_ctx *c) { return c->conic_opacity; }
const float *gsro_cov3D(const gsro_ctx *c) { return c->cov3D; }
const float *gsro_rgb(const gsro_ctx *c) { return c->rgb; }
const int32_t *gsro_rect(const gsro_ctx *c) { return c->rect; }
const uint32_t *gsro_tiles_touched(const gsro_ctx *c) { return c->tiles_touched; }
const uint32_t *gsro_offsets(const gsro_ctx *c) { return c->offsets; }
const uint64_t *gsro_keys(const gsro_ctx *c) { return c->keys; }
const uint32_t *gsro_point_list(const gsro_ctx *c) { return c->point_list; }
const uint32_t *gsro_ranges(const gsro_ctx *c) { return c->ranges; }
const float *gsro_final_T(const gsro_ctx *c) { return c->final_T; }
const uint32_t *gsro_n_contrib(const gsro_ctx *c) { return c->n_contrib; }
const uint8_t *gsro_ambiguous(const gsro_ctx *c) { return c->ambiguous; }

/* A.1-1 only: boolean visibility (mark_visible) */
void gsro_mark_visible(const float *view, int P, const float *means3D, uint8_t *present) {
  for (int i = 0; i < P; ++i) {
    const float *p = means3D + 3 * i;
    float z = view[2] * p[0] + view[6] * p[1] + view[10] * p[2] + view[14];
    present[i] = z > NEAR_Z;
  }
}
