/*
 * oracle/raster_ref.c -- CPU restatement of the reference curve-Gaussian rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle (and the "port" CPU baseline
 * timed by bench.py); nothing under curve_gaussian_amd/ may import, link or call it.
 *
 * PARITY UNPINNED: the reference rasterizer (submodules/diff-cur-rasterization) ships no tests or
 * golden vectors and cannot be compiled here (needs nvcc + cub + the un-vendored GLM submodule), so
 * this restatement is pinned only by (i) line-by-line derivation from the sources cited below,
 * (ii) an independent PyTorch-autograd forward (oracle/torch_ref.py) and (iii) finite differences
 * (tests/test_oracle_raster.py).
 *
 * Reference files restated (paths under /root/reference/submodules/diff-cur-rasterization/):
 *   cuda_rasterizer/auxiliary.h      ndc2Pix :40-43, getRect :45-55, transformPoint* :70-109, in_frustum :151-176
 *   cuda_rasterizer/forward.cu       computeColorFromSH :20-75, computeCov2D :78-113, computeCov3D :118-152,
 *                                    preprocessCUDA :155-274, renderCUDA :279-417
 *   cuda_rasterizer/rasterizer_impl.cu  getHigherMsb :35-50, duplicateWithKeys :70-111, identifyTileRanges :116-138,
 *                                    forward host sequence :198-347, backward host sequence :351-466
 *   cuda_rasterizer/backward.cu      SH bwd :23-141, computeCov2DCUDA :146-325, computeCov3D bwd :329-392,
 *                                    preprocessCUDA bwd :397-448, renderCUDA bwd :451-675
 *   rasterize_points.cu              output / gradient allocation and zero-init :35-130, :132-239
 *
 * GLM (missing third-party submodule, no pinned commit) is restated by hand: glm::mat3(a..i) fills
 * COLUMNS, X[i][j] is column i row j, A*B is the math product.
 *
 * Arithmetic is float32 wherever the reference's is.  The reference accumulates per-splat gradients
 * with float atomicAdd in an undefined order; the oracle accumulates those sums in double and rounds
 * once, i.e. it returns the centre of the reference's own run-to-run distribution.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16 /* cuda_rasterizer/config.h:17-18 */
#define BLOCK_Y 16
#define NUM_CHANNELS 1 /* config.h:15 */
#define NUM_ALL_MAP 4  /* config.h:16 */

static const float SH_C0 = 0.28209479177387814f; /* auxiliary.h:21-38 */
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct {
    /* problem */
    int P, D, M, W, H;
    int grid_x, grid_y;
    /* per-splat state (GeometryState, rasterizer_impl.cu:155-170) */
    float* depths;        /* P */
    uint8_t* clamped;     /* P (single channel, forward.cu:73) */
    int* radii;           /* P */
    float* means2D;       /* 2P */
    float* cov3D;         /* 6P */
    float* conic_opacity; /* 4P */
    float* rgb;           /* P */
    uint32_t* tiles_touched; /* P */
    uint32_t* point_offsets; /* P */
    /* binning state */
    int64_t R;
    uint64_t* keys;       /* R sorted */
    uint32_t* point_list; /* R sorted */
    /* image state */
    uint32_t* ranges;   /* 2*tiles */
    float* final_T;     /* H*W */
    uint32_t* n_contrib; /* H*W */
} ora_ctx;

static float fminf_(float a, float b) { return a < b ? a : b; }
static float fmaxf_(float a, float b) { return a > b ? a : b; }
static int imin_(int a, int b) { return a < b ? a : b; }
static int imax_(int a, int b) { return a > b ? a : b; }

/* auxiliary.h:40-43 -- evaluated in double because of the 1.0 / 0.5 literals, then narrowed */
static float ndc2Pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

/* auxiliary.h:45-55 */
static void getRect(float px, float py, int max_radius, int gx, int gy, uint32_t* rmin, uint32_t* rmax) {
    rmin[0] = (uint32_t)imin_(gx, imax_(0, (int)((px - (float)max_radius) / (float)BLOCK_X)));
    rmin[1] = (uint32_t)imin_(gy, imax_(0, (int)((py - (float)max_radius) / (float)BLOCK_Y)));
    rmax[0] = (uint32_t)imin_(gx, imax_(0, (int)((px + (float)max_radius + (float)(BLOCK_X - 1)) / (float)BLOCK_X)));
    rmax[1] = (uint32_t)imin_(gy, imax_(0, (int)((py + (float)max_radius + (float)(BLOCK_Y - 1)) / (float)BLOCK_Y)));
}

/* auxiliary.h:70-89 */
static void transformPoint4x3(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void transformPoint4x4(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* rasterizer_impl.cu:35-50 (kept for the key-width bookkeeping reported by bench/DESIGN) */
uint32_t ora_get_higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* forward.cu:20-75, single channel */
static float sh_forward(int idx, int deg, int max_coeffs, const float* means, const float* campos, const float* shs,
                        uint8_t* clamped) {
    float dir[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] /= len; dir[1] /= len; dir[2] /= len;
    const float* sh = shs + (size_t)idx * max_coeffs;
    float result = SH_C0 * sh[0];
    if (deg > 0) {
        float x = dir[0], y = dir[1], z = dir[2];
        result = result - SH_C1 * y * sh[1] + SH_C1 * z * sh[2] - SH_C1 * x * sh[3];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            result = result + SH_C2[0] * xy * sh[4] + SH_C2[1] * yz * sh[5] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[6] +
                     SH_C2[3] * xz * sh[7] + SH_C2[4] * (xx - yy) * sh[8];
            if (deg > 2) {
                result = result + SH_C3[0] * y * (3.0f * xx - yy) * sh[9] + SH_C3[1] * xy * z * sh[10] +
                         SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11] +
                         SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12] +
                         SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13] + SH_C3[5] * z * (xx - yy) * sh[14] +
                         SH_C3[6] * x * (xx - 3.0f * yy) * sh[15];
            }
        }
    }
    result += 0.5f;
    clamped[idx] = (result < 0);
    return fmaxf_(result, 0.0f);
}

/* forward.cu:118-152.  Rq = rotation of the UN-normalised quaternion (r,x,y,z); Sigma = Rq S^2 Rq^T. */
static void quat_to_Rq(const float* q, float Rq[3][3]) {
    float r = q[0], x = q[1], y = q[2], z = q[3];
    Rq[0][0] = 1.f - 2.f * (y * y + z * z); Rq[0][1] = 2.f * (x * y - r * z); Rq[0][2] = 2.f * (x * z + r * y);
    Rq[1][0] = 2.f * (x * y + r * z); Rq[1][1] = 1.f - 2.f * (x * x + z * z); Rq[1][2] = 2.f * (y * z - r * x);
    Rq[2][0] = 2.f * (x * z - r * y); Rq[2][1] = 2.f * (y * z + r * x); Rq[2][2] = 1.f - 2.f * (x * x + y * y);
}
static void computeCov3D(const float* scale, float mod, const float* rot, float* cov3D) {
    float Rq[3][3];
    quat_to_Rq(rot, Rq);
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    /* M = S * R_glm, math M[k][a] = s_k * Rq[a][k];  Sigma = M^T M */
    float Mm[3][3];
    for (int k = 0; k < 3; k++)
        for (int a = 0; a < 3; a++) Mm[k][a] = s[k] * Rq[a][k];
    float Sg[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Sg[a][b] = Mm[0][a] * Mm[0][b] + Mm[1][a] * Mm[1][b] + Mm[2][a] * Mm[2][b];
    cov3D[0] = Sg[0][0]; cov3D[1] = Sg[0][1]; cov3D[2] = Sg[0][2];
    cov3D[3] = Sg[1][1]; cov3D[4] = Sg[1][2]; cov3D[5] = Sg[2][2];
}

/* Shared by forward.cu:78-113 and backward.cu:172-200: t (clamped), Mt = Jm*Rwc (rows 0,1), cov2D (a,b,c). */
static void cov2d_terms(const float* mean, float fx, float fy, float tan_fovx, float tan_fovy, const float* cov3D,
                        const float* vm, float t[3], float Mt[2][3], float cov[3], float* txtz_out, float* tytz_out) {
    transformPoint4x3(mean, vm, t);
    const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    if (txtz_out) { *txtz_out = txtz; *tytz_out = tytz; }
    /* Jm rows */
    const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* Rwc[r][c] = vm[c*4+r] */
    for (int j = 0; j < 3; j++) {
        /* glm T = W*J: T[col i][row j] = sum_k W[k][j]*J[i][k]; the k=1 (resp. k=0) term is an exact 0*x */
        Mt[0][j] = vm[j * 4 + 0] * J00 + vm[j * 4 + 1] * 0.0f + vm[j * 4 + 2] * J02;
        Mt[1][j] = vm[j * 4 + 0] * 0.0f + vm[j * 4 + 1] * J11 + vm[j * 4 + 2] * J12;
    }
    const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
    float U[2][3]; /* U = Mt * V */
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) U[i][j] = Mt[i][0] * V[0][j] + Mt[i][1] * V[1][j] + Mt[i][2] * V[2][j];
    cov[0] = U[0][0] * Mt[0][0] + U[0][1] * Mt[0][1] + U[0][2] * Mt[0][2];
    cov[1] = U[0][0] * Mt[1][0] + U[0][1] * Mt[1][1] + U[0][2] * Mt[1][2];
    cov[2] = U[1][0] * Mt[1][0] + U[1][1] * Mt[1][1] + U[1][2] * Mt[1][2];
}

typedef struct { uint64_t key; uint32_t val; uint32_t seq; } kv_t;
static int kv_cmp(const void* a, const void* b) {
    const kv_t* x = (const kv_t*)a; const kv_t* y = (const kv_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable: cub radix sort keeps emission order */
}

void ora_free(ora_ctx* c) {
    if (!c) return;
    free(c->depths); free(c->clamped); free(c->radii); free(c->means2D); free(c->cov3D); free(c->conic_opacity);
    free(c->rgb); free(c->tiles_touched); free(c->point_offsets); free(c->keys); free(c->point_list);
    free(c->ranges); free(c->final_T); free(c->n_contrib);
    free(c);
}

/*
 * Forward: rasterizer_impl.cu:198-347.  Null pointers play the role of the reference's empty tensors
 * (shs / colors_precomp / scales / rotations / cov3D_precomp).  Outputs are caller-owned:
 *   out_color [1,H,W], out_invdepth [1,H,W], out_all_map [4,H,W], radii [P]  (all pre-zeroed here, as
 *   rasterize_points.cu:71-79 does).
 * Returns a context that holds geom/binning/image state for the backward (NULL on allocation failure).
 */
ora_ctx* ora_forward(int P, int D, int M, const float* background, int W, int H, const float* means3D,
                     const float* shs, const float* colors_precomp, const float* opacities, const float* scales,
                     float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* all_map,
                     const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                     float tan_fovy, int prefiltered, int antialiasing, int render_geo, float* out_color,
                     float* out_invdepth, float* out_all_map, int* radii_out) {
    ora_ctx* c = (ora_ctx*)calloc(1, sizeof(ora_ctx));
    if (!c) return NULL;
    c->P = P; c->D = D; c->M = M; c->W = W; c->H = H;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    c->grid_x = gx; c->grid_y = gy;
    const size_t NP = (size_t)(P > 0 ? P : 1), NPIX = (size_t)W * H, NT = (size_t)gx * gy;
    c->depths = (float*)calloc(NP, 4); c->clamped = (uint8_t*)calloc(NP, 1); c->radii = (int*)calloc(NP, 4);
    c->means2D = (float*)calloc(2 * NP, 4); c->cov3D = (float*)calloc(6 * NP, 4);
    c->conic_opacity = (float*)calloc(4 * NP, 4); c->rgb = (float*)calloc(NP, 4);
    c->tiles_touched = (uint32_t*)calloc(NP, 4); c->point_offsets = (uint32_t*)calloc(NP, 4);
    c->ranges = (uint32_t*)calloc(2 * NT + 2, 4); c->final_T = (float*)calloc(NPIX + 1, 4);
    c->n_contrib = (uint32_t*)calloc(NPIX + 1, 4);
    memset(out_color, 0, NPIX * NUM_CHANNELS * 4);
    memset(out_invdepth, 0, NPIX * 4);
    memset(out_all_map, 0, NPIX * NUM_ALL_MAP * 4);
    if (P == 0) return c; /* rasterize_points.cu:91 */
    (void)prefiltered;

    const float focal_y = (float)H / (2.0f * tan_fovy); /* rasterizer_impl.cu:227-228 */
    const float focal_x = (float)W / (2.0f * tan_fovx);

    /* K1 preprocess, forward.cu:155-274 */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        c->radii[idx] = 0;
        c->tiles_touched[idx] = 0;
        const float* p_orig = means3D + 3 * (size_t)idx;
        float p_view[3];
        transformPoint4x3(p_orig, viewmatrix, p_view);
        if (p_view[2] <= 0.2f) continue; /* auxiliary.h:166 */
        float p_hom[4];
        transformPoint4x4(p_orig, projmatrix, p_hom);
        float p_w = 1.0f / (p_hom[3] + 0.0000001f);
        float p_proj[3] = {p_hom[0] * p_w, p_hom[1] * p_w, p_hom[2] * p_w};
        const float* cov3D;
        if (cov3D_precomp) {
            cov3D = cov3D_precomp + 6 * (size_t)idx;
        } else {
            computeCov3D(scales + 3 * (size_t)idx, scale_modifier, rotations + 4 * (size_t)idx, c->cov3D + 6 * (size_t)idx);
            cov3D = c->cov3D + 6 * (size_t)idx;
        }
        float t[3], Mt[2][3], cov[3];
        cov2d_terms(p_orig, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, t, Mt, cov, NULL, NULL);
        const float h_var = 0.3f;
        const float det_cov = cov[0] * cov[2] - cov[1] * cov[1];
        cov[0] += h_var; cov[2] += h_var;
        const float det_cov_plus_h_cov = cov[0] * cov[2] - cov[1] * cov[1];
        float h_convolution_scaling = 1.0f;
        if (antialiasing) h_convolution_scaling = sqrtf(fmaxf_(0.000025f, det_cov / det_cov_plus_h_cov));
        const float det = det_cov_plus_h_cov;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        float mid = 0.5f * (cov[0] + cov[2]);
        float lambda1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lambda2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lambda1, lambda2)));
        float px = ndc2Pix(p_proj[0], W), py = ndc2Pix(p_proj[1], H);
        uint32_t rmin[2], rmax[2];
        getRect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) c->rgb[idx] = sh_forward(idx, D, M, means3D, campos, shs, c->clamped);
        c->depths[idx] = p_view[2];
        c->radii[idx] = (int)my_radius;
        c->means2D[2 * idx] = px; c->means2D[2 * idx + 1] = py;
        c->conic_opacity[4 * idx + 0] = conic[0]; c->conic_opacity[4 * idx + 1] = conic[1];
        c->conic_opacity[4 * idx + 2] = conic[2];
        c->conic_opacity[4 * idx + 3] = opacities[idx] * h_convolution_scaling;
        c->tiles_touched[idx] = (rmax[1] - rmin[1]) * (rmax[0] - rmin[0]);
    }
    memcpy(radii_out, c->radii, (size_t)P * 4);

    /* K2 inclusive scan, rasterizer_impl.cu:283 */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += c->tiles_touched[i]; c->point_offsets[i] = acc; }
    const int64_t R = acc;
    c->R = R;

    /* K3 duplicateWithKeys :70-111, K4 stable sort :309-314 */
    kv_t* kv = (kv_t*)malloc((size_t)(R > 0 ? R : 1) * sizeof(kv_t));
#pragma omp parallel for schedule(dynamic, 1024)
    for (int idx = 0; idx < P; idx++) {
        if (c->radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : c->point_offsets[idx - 1];
            uint32_t rmin[2], rmax[2];
            getRect(c->means2D[2 * idx], c->means2D[2 * idx + 1], c->radii[idx], gx, gy, rmin, rmax);
            uint32_t dbits;
            memcpy(&dbits, &c->depths[idx], 4);
            for (uint32_t y = rmin[1]; y < rmax[1]; y++)
                for (uint32_t x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                    key <<= 32; key |= dbits;
                    kv[off].key = key; kv[off].val = (uint32_t)idx; kv[off].seq = off;
                    off++;
                }
        }
    }
    qsort(kv, (size_t)R, sizeof(kv_t), kv_cmp);
    c->keys = (uint64_t*)malloc((size_t)(R > 0 ? R : 1) * 8);
    c->point_list = (uint32_t*)malloc((size_t)(R > 0 ? R : 1) * 4);
    for (int64_t i = 0; i < R; i++) { c->keys[i] = kv[i].key; c->point_list[i] = kv[i].val; }
    free(kv);

    /* K5 identifyTileRanges :116-138 (ranges zeroed by calloc = cudaMemset :316) */
    for (int64_t i = 0; i < R; i++) {
        uint32_t currtile = (uint32_t)(c->keys[i] >> 32);
        if (i == 0) c->ranges[2 * currtile] = 0;
        else {
            uint32_t prevtile = (uint32_t)(c->keys[i - 1] >> 32);
            if (currtile != prevtile) { c->ranges[2 * prevtile + 1] = (uint32_t)i; c->ranges[2 * currtile] = (uint32_t)i; }
        }
        if (i == R - 1) c->ranges[2 * currtile + 1] = (uint32_t)R;
    }

    /* K6 renderCUDA forward.cu:279-417, one pixel at a time (block-level voting does not change per-pixel results) */
    const float* features = colors_precomp ? colors_precomp : c->rgb;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const size_t pix_id = (size_t)W * pyi + pxi;
                const float pixfx = (float)pxi, pixfy = (float)pyi;
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float C[NUM_CHANNELS] = {0};
                float Amap[NUM_ALL_MAP] = {0};
                float expected_invdepth = 0.0f;
                for (uint32_t k = r0; k < r1; k++) {
                    contributor++;
                    const uint32_t id = c->point_list[k];
                    const float dx = c->means2D[2 * id] - pixfx, dy = c->means2D[2 * id + 1] - pixfy;
                    const float* co = c->conic_opacity + 4 * (size_t)id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf_(0.99f, co[3] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done = true */
                    for (int ch = 0; ch < NUM_CHANNELS; ch++) C[ch] += features[id * NUM_CHANNELS + ch] * alpha * T;
                    expected_invdepth += (1 / c->depths[id]) * alpha * T;
                    if (render_geo)
                        for (int ch = 0; ch < NUM_ALL_MAP; ch++) Amap[ch] += all_map[id * NUM_ALL_MAP + ch] * alpha * T;
                    T = test_T;
                    last_contributor = contributor;
                }
                c->final_T[pix_id] = T;
                c->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < NUM_CHANNELS; ch++) out_color[(size_t)ch * H * W + pix_id] = C[ch] + T * background[ch];
                out_invdepth[pix_id] = expected_invdepth;
                if (render_geo)
                    for (int ch = 0; ch < NUM_ALL_MAP; ch++) out_all_map[(size_t)ch * H * W + pix_id] = Amap[ch];
            }
    }
    return c;
}

/* accessors (ctypes-friendly) */
int64_t ora_num_rendered(const ora_ctx* c) { return c->R; }
const float* ora_means2D(const ora_ctx* c) { return c->means2D; }
const float* ora_depths(const ora_ctx* c) { return c->depths; }
const float* ora_cov3D(const ora_ctx* c) { return c->cov3D; }
const float* ora_conic_opacity(const ora_ctx* c) { return c->conic_opacity; }
const uint32_t* ora_tiles_touched(const ora_ctx* c) { return c->tiles_touched; }
const uint32_t* ora_point_list(const ora_ctx* c) { return c->point_list; }
const uint64_t* ora_keys(const ora_ctx* c) { return c->keys; }
const uint32_t* ora_ranges(const ora_ctx* c) { return c->ranges; }
const float* ora_final_T(const ora_ctx* c) { return c->final_T; }
const uint32_t* ora_n_contrib(const ora_ctx* c) { return c->n_contrib; }
const float* ora_rgb(const ora_ctx* c) { return c->rgb; }

/* markVisible: rasterizer_impl.cu:54-66,141-153 */
void ora_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present) {
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        float pv[3];
        transformPoint4x3(means3D + 3 * (size_t)i, viewmatrix, pv);
        present[i] = pv[2] > 0.2f;
    }
}

/* backward.cu:23-141 single channel; accumulates into dL_dmeans (+=) and writes dL_dsh */
static void sh_backward(int idx, int deg, int max_coeffs, const float* means, const float* campos, const float* shs,
                        const uint8_t* clamped, const float* dL_dcolor, float* dL_dmeans, float* dL_dshs) {
    float dir_orig[3] = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    float x = dir_orig[0] / len, y = dir_orig[1] / len, z = dir_orig[2] / len;
    const float* sh = shs + (size_t)idx * max_coeffs;
    float dL_dRGB = dL_dcolor[idx];
    dL_dRGB *= clamped[idx] ? 0 : 1;
    float dRGBdx = 0, dRGBdy = 0, dRGBdz = 0;
    float* dL_dsh = dL_dshs + (size_t)idx * max_coeffs;
    dL_dsh[0] = SH_C0 * dL_dRGB;
    if (deg > 0) {
        dL_dsh[1] = (-SH_C1 * y) * dL_dRGB; dL_dsh[2] = (SH_C1 * z) * dL_dRGB; dL_dsh[3] = (-SH_C1 * x) * dL_dRGB;
        dRGBdx = -SH_C1 * sh[3]; dRGBdy = -SH_C1 * sh[1]; dRGBdz = SH_C1 * sh[2];
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dL_dsh[4] = (SH_C2[0] * xy) * dL_dRGB; dL_dsh[5] = (SH_C2[1] * yz) * dL_dRGB;
            dL_dsh[6] = (SH_C2[2] * (2.f * zz - xx - yy)) * dL_dRGB; dL_dsh[7] = (SH_C2[3] * xz) * dL_dRGB;
            dL_dsh[8] = (SH_C2[4] * (xx - yy)) * dL_dRGB;
            dRGBdx += SH_C2[0] * y * sh[4] + SH_C2[2] * 2.f * -x * sh[6] + SH_C2[3] * z * sh[7] + SH_C2[4] * 2.f * x * sh[8];
            dRGBdy += SH_C2[0] * x * sh[4] + SH_C2[1] * z * sh[5] + SH_C2[2] * 2.f * -y * sh[6] + SH_C2[4] * 2.f * -y * sh[8];
            dRGBdz += SH_C2[1] * y * sh[5] + SH_C2[2] * 2.f * 2.f * z * sh[6] + SH_C2[3] * x * sh[7];
            if (deg > 2) {
                dL_dsh[9] = (SH_C3[0] * y * (3.f * xx - yy)) * dL_dRGB; dL_dsh[10] = (SH_C3[1] * xy * z) * dL_dRGB;
                dL_dsh[11] = (SH_C3[2] * y * (4.f * zz - xx - yy)) * dL_dRGB;
                dL_dsh[12] = (SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dL_dRGB;
                dL_dsh[13] = (SH_C3[4] * x * (4.f * zz - xx - yy)) * dL_dRGB;
                dL_dsh[14] = (SH_C3[5] * z * (xx - yy)) * dL_dRGB; dL_dsh[15] = (SH_C3[6] * x * (xx - 3.f * yy)) * dL_dRGB;
                dRGBdx += (SH_C3[0] * sh[9] * 3.f * 2.f * xy + SH_C3[1] * sh[10] * yz + SH_C3[2] * sh[11] * -2.f * xy +
                           SH_C3[3] * sh[12] * -3.f * 2.f * xz + SH_C3[4] * sh[13] * (-3.f * xx + 4.f * zz - yy) +
                           SH_C3[5] * sh[14] * 2.f * xz + SH_C3[6] * sh[15] * 3.f * (xx - yy));
                dRGBdy += (SH_C3[0] * sh[9] * 3.f * (xx - yy) + SH_C3[1] * sh[10] * xz +
                           SH_C3[2] * sh[11] * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * sh[12] * -3.f * 2.f * yz +
                           SH_C3[4] * sh[13] * -2.f * xy + SH_C3[5] * sh[14] * -2.f * yz + SH_C3[6] * sh[15] * -3.f * 2.f * xy);
                dRGBdz += (SH_C3[1] * sh[10] * xy + SH_C3[2] * sh[11] * 4.f * 2.f * yz +
                           SH_C3[3] * sh[12] * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * sh[13] * 4.f * 2.f * xz +
                           SH_C3[5] * sh[14] * (xx - yy));
            }
        }
    }
    float ddir[3] = {dRGBdx * dL_dRGB, dRGBdy * dL_dRGB, dRGBdz * dL_dRGB};
    /* dnormvdv auxiliary.h:119-129 */
    float vx = dir_orig[0], vy = dir_orig[1], vz = dir_orig[2];
    float sum2 = vx * vx + vy * vy + vz * vz;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dL_dmeans[3 * idx + 0] += ((+sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * invsum32;
    dL_dmeans[3 * idx + 2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * invsum32;
}

static inline void atomic_add_d(double* p, double v) {
#pragma omp atomic
    *p += v;
}

/*
 * Backward: rasterizer_impl.cu:351-466 with the allocation/zero-init of rasterize_points.cu:173-193.
 * dL_dout_invdepth may be NULL (reference: grad tensor with size(0)==0).  All gradient outputs are
 * caller-owned and zeroed here:  dL_dmeans2D [P,3], dL_dcolors [P,1], dL_dopacity [P,1], dL_dmeans3D [P,3],
 * dL_dcov3D [P,6], dL_dsh [P,M] (single channel, see SURVEY quirk 16 -- the reference allocates [P,M,3] but
 * writes only the first P*M floats; callers that want the reference's tensor shape pad), dL_dscales [P,3],
 * dL_drotations [P,4], dL_dall_map [P,4], dL_dconic [P,4] (scratch, exposed for tests), dL_dinvdepths [P].
 */
void ora_backward(const ora_ctx* c, const float* background, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* all_maps, const float* opacities, const float* scales,
                  float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                  const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const float* dL_dpix,
                  const float* dL_dout_invdepth, const float* dL_dout_all_map, int antialiasing, int render_geo,
                  float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D,
                  float* dL_dsh, float* dL_dscales, float* dL_drotations, float* dL_dall_map, float* dL_dconic,
                  float* dL_dinvdepths) {
    const int P = c->P, W = c->W, H = c->H, gx = c->grid_x, gy = c->grid_y, D = c->D, M = c->M;
    const size_t NP = (size_t)P;
    memset(dL_dmeans2D, 0, NP * 3 * 4); memset(dL_dcolors, 0, NP * 4); memset(dL_dopacity, 0, NP * 4);
    memset(dL_dmeans3D, 0, NP * 3 * 4); memset(dL_dcov3D, 0, NP * 6 * 4);
    if (M > 0) memset(dL_dsh, 0, NP * (size_t)M * 4);
    memset(dL_dscales, 0, NP * 3 * 4); memset(dL_drotations, 0, NP * 4 * 4); memset(dL_dall_map, 0, NP * 4 * 4);
    memset(dL_dconic, 0, NP * 4 * 4);
    if (dL_dinvdepths) memset(dL_dinvdepths, 0, NP * 4);
    if (P == 0) return;
    const float focal_y = (float)H / (2.0f * tan_fovy);
    const float focal_x = (float)W / (2.0f * tan_fovx);
    const float* colors = colors_precomp ? colors_precomp : c->rgb;

    /* double accumulators for the atomicAdd targets of backward.cu:613-672 */
    double* a_mean2D = (double*)calloc(2 * NP, 8);
    double* a_conic = (double*)calloc(3 * NP, 8);
    double* a_opac = (double*)calloc(NP, 8);
    double* a_color = (double*)calloc(NP, 8);
    double* a_invd = (double*)calloc(NP, 8);
    double* a_amap = (double*)calloc(4 * NP, 8);

    /* K8 renderCUDA backward.cu:451-675 */
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < gx * gy; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = c->ranges[2 * tile], r1 = c->ranges[2 * tile + 1];
        if (r1 == r0) continue;
        for (int ly = 0; ly < BLOCK_Y; ly++)
            for (int lx = 0; lx < BLOCK_X; lx++) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const size_t pix_id = (size_t)W * pyi + pxi;
                const float pixfx = (float)pxi, pixfy = (float)pyi;
                const float T_final = c->final_T[pix_id];
                float T = T_final;
                uint32_t contributor = r1 - r0;
                const uint32_t last_contributor = c->n_contrib[pix_id];
                float accum_rec[NUM_CHANNELS] = {0};
                float dL_dpixel[NUM_CHANNELS];
                float dL_invdepth = 0.f;
                float accum_invdepth_rec = 0;
                float accum_all_map[NUM_ALL_MAP] = {0};
                float dL_dout_amap[NUM_ALL_MAP] = {0};
                for (int i = 0; i < NUM_CHANNELS; i++) dL_dpixel[i] = dL_dpix[(size_t)i * H * W + pix_id];
                if (dL_dout_invdepth) dL_invdepth = dL_dout_invdepth[pix_id];
                if (render_geo)
                    for (int i = 0; i < NUM_ALL_MAP; i++) dL_dout_amap[i] = dL_dout_all_map[(size_t)i * H * W + pix_id];
                float last_alpha = 0;
                float last_color[NUM_CHANNELS] = {0};
                float last_invdepth = 0;
                float last_all_map[NUM_ALL_MAP] = {0};
                const float ddelx_dx = (float)(0.5 * W); /* backward.cu:542-543 (double product narrowed to float) */
                const float ddely_dy = (float)(0.5 * H);
                for (uint32_t k = r1; k-- > r0;) {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t id = c->point_list[k];
                    const float dx = c->means2D[2 * id] - pixfx, dy = c->means2D[2 * id + 1] - pixfy;
                    const float* co = c->conic_opacity + 4 * (size_t)id;
                    const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0.0f) continue;
                    const float G = expf(power);
                    const float alpha = fminf_(0.99f, co[3] * G);
                    if (alpha < 1.0f / 255.0f) continue;
                    T = T / (1.f - alpha);
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    for (int ch = 0; ch < NUM_CHANNELS; ch++) {
                        const float col = colors[id * NUM_CHANNELS + ch];
                        accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                        last_color[ch] = col;
                        const float dL_dchannel = dL_dpixel[ch];
                        dL_dalpha += (col - accum_rec[ch]) * dL_dchannel;
                        atomic_add_d(&a_color[id * NUM_CHANNELS + ch], (double)(dchannel_dcolor * dL_dchannel));
                    }
                    if (dL_dinvdepths) { /* backward.cu:617-624 */
                        const float invd = 1.f / c->depths[id];
                        accum_invdepth_rec = last_alpha * last_invdepth + (1.f - last_alpha) * accum_invdepth_rec;
                        last_invdepth = invd;
                        dL_dalpha += (invd - accum_invdepth_rec) * dL_invdepth;
                        atomic_add_d(&a_invd[id], (double)(dchannel_dcolor * dL_invdepth));
                    }
                    if (render_geo) {
                        for (int ch = 0; ch < NUM_ALL_MAP; ch++) {
                            const float col = all_maps[id * NUM_ALL_MAP + ch];
                            accum_all_map[ch] = last_alpha * last_all_map[ch] + (1.f - last_alpha) * accum_all_map[ch];
                            last_all_map[ch] = col;
                            const float dL_dchannel = dL_dout_amap[ch];
                            dL_dalpha += (col - accum_all_map[ch]) * dL_dchannel;
                            atomic_add_d(&a_amap[id * NUM_ALL_MAP + ch], (double)(dchannel_dcolor * dL_dchannel));
                        }
                    }
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    float bg_dot_dpixel = 0;
                    for (int i = 0; i < NUM_CHANNELS; i++) bg_dot_dpixel += background[i] * dL_dpixel[i];
                    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                    const float dL_dG = co[3] * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                    const float dG_ddely = -gdy * co[2] - gdx * co[1];
                    atomic_add_d(&a_mean2D[2 * id + 0], (double)(dL_dG * dG_ddelx * ddelx_dx));
                    atomic_add_d(&a_mean2D[2 * id + 1], (double)(dL_dG * dG_ddely * ddely_dy));
                    atomic_add_d(&a_conic[3 * id + 0], (double)(-0.5f * gdx * dx * dL_dG));
                    atomic_add_d(&a_conic[3 * id + 1], (double)(-0.5f * gdx * dy * dL_dG));
                    atomic_add_d(&a_conic[3 * id + 2], (double)(-0.5f * gdy * dy * dL_dG));
                    atomic_add_d(&a_opac[id], (double)(G * dL_dalpha));
                }
            }
    }
    for (size_t i = 0; i < NP; i++) {
        dL_dmeans2D[3 * i + 0] = (float)a_mean2D[2 * i + 0];
        dL_dmeans2D[3 * i + 1] = (float)a_mean2D[2 * i + 1]; /* .z stays 0, backward.cu:663-664 */
        dL_dconic[4 * i + 0] = (float)a_conic[3 * i + 0];
        dL_dconic[4 * i + 1] = (float)a_conic[3 * i + 1];
        dL_dconic[4 * i + 3] = (float)a_conic[3 * i + 2]; /* float4 .w, backward.cu:669 */
        dL_dopacity[i] = (float)a_opac[i];
        dL_dcolors[i] = (float)a_color[i];
        if (dL_dinvdepths) dL_dinvdepths[i] = (float)a_invd[i];
        for (int ch = 0; ch < 4; ch++) dL_dall_map[4 * i + ch] = (float)a_amap[4 * i + ch];
    }
    free(a_mean2D); free(a_conic); free(a_opac); free(a_color); free(a_invd); free(a_amap);

    /* K9 computeCov2DCUDA backward.cu:146-325, then K10 preprocessCUDA :397-448 */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(c->radii[idx] > 0)) continue;
        const float* cov3D = (cov3D_precomp ? cov3D_precomp : c->cov3D) + 6 * (size_t)idx;
        const float* mean = means3D + 3 * (size_t)idx;
        const float dcx = dL_dconic[4 * idx], dcy = dL_dconic[4 * idx + 1], dcz = dL_dconic[4 * idx + 3];
        float t[3], T_[2][3], cov[3], txtz, tytz;
        cov2d_terms(mean, focal_x, focal_y, tan_fovx, tan_fovy, cov3D, viewmatrix, t, T_, cov, &txtz, &tytz);
        const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
        const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        float c_xx = cov[0], c_xy = cov[1], c_yy = cov[2];
        const float h_var = 0.3f;
        float d_inside_root = 0.f;
        if (antialiasing) {
            const float det_cov = c_xx * c_yy - c_xy * c_xy;
            c_xx += h_var; c_yy += h_var;
            const float det_cov_plus_h_cov = c_xx * c_yy - c_xy * c_xy;
            const float h_convolution_scaling = sqrtf(fmaxf_(0.000025f, det_cov / det_cov_plus_h_cov));
            const float dL_dopacity_v = dL_dopacity[idx];
            const float d_h_convolution_scaling = dL_dopacity_v * opacities[idx];
            dL_dopacity[idx] = dL_dopacity_v * h_convolution_scaling;
            d_inside_root = (det_cov / det_cov_plus_h_cov) <= 0.000025f ? 0.f : d_h_convolution_scaling / (2 * h_convolution_scaling);
        } else {
            c_xx += h_var; c_yy += h_var;
        }
        float dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
        if (antialiasing) {
            const float x = c_xx, y = c_yy, z = c_xy, w = h_var;
            const float sqv = (w * w + w * (x + y) + x * y - z * z);
            const float denom_f = d_inside_root / (sqv * sqv);
            dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
            dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
            dL_dc_xy = -2.f * w * z * (w + x + y) * denom_f;
        }
        float denom = c_xx * c_yy - c_xy * c_xy;
        float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov3D + 6 * (size_t)idx;
        if (denom2inv != 0) {
            dL_dc_xx += denom2inv * (-c_yy * c_yy * dcx + 2 * c_xy * c_yy * dcy + (denom - c_xx * c_yy) * dcz);
            dL_dc_yy += denom2inv * (-c_xx * c_xx * dcz + 2 * c_xx * c_xy * dcy + (denom - c_xx * c_yy) * dcx);
            dL_dc_xy += denom2inv * 2 * (c_xy * c_yy * dcx - (denom + 2 * c_xy * c_xy) * dcy + c_xx * c_xy * dcz);
            dcov[0] = (T_[0][0] * T_[0][0] * dL_dc_xx + T_[0][0] * T_[1][0] * dL_dc_xy + T_[1][0] * T_[1][0] * dL_dc_yy);
            dcov[3] = (T_[0][1] * T_[0][1] * dL_dc_xx + T_[0][1] * T_[1][1] * dL_dc_xy + T_[1][1] * T_[1][1] * dL_dc_yy);
            dcov[5] = (T_[0][2] * T_[0][2] * dL_dc_xx + T_[0][2] * T_[1][2] * dL_dc_xy + T_[1][2] * T_[1][2] * dL_dc_yy);
            dcov[1] = 2 * T_[0][0] * T_[0][1] * dL_dc_xx + (T_[0][0] * T_[1][1] + T_[0][1] * T_[1][0]) * dL_dc_xy + 2 * T_[1][0] * T_[1][1] * dL_dc_yy;
            dcov[2] = 2 * T_[0][0] * T_[0][2] * dL_dc_xx + (T_[0][0] * T_[1][2] + T_[0][2] * T_[1][0]) * dL_dc_xy + 2 * T_[1][0] * T_[1][2] * dL_dc_yy;
            dcov[4] = 2 * T_[0][2] * T_[0][1] * dL_dc_xx + (T_[0][1] * T_[1][2] + T_[0][2] * T_[1][1]) * dL_dc_xy + 2 * T_[1][1] * T_[1][2] * dL_dc_yy;
        } else {
            for (int i = 0; i < 6; i++) dcov[i] = 0;
        }
        const float V[3][3] = {{cov3D[0], cov3D[1], cov3D[2]}, {cov3D[1], cov3D[3], cov3D[4]}, {cov3D[2], cov3D[4], cov3D[5]}};
        float dT0[3], dT1[3];
        for (int j = 0; j < 3; j++) {
            const float u0 = T_[0][0] * V[j][0] + T_[0][1] * V[j][1] + T_[0][2] * V[j][2];
            const float u1 = T_[1][0] * V[j][0] + T_[1][1] * V[j][1] + T_[1][2] * V[j][2];
            dT0[j] = 2 * u0 * dL_dc_xx + u1 * dL_dc_xy;
            dT1[j] = 2 * u1 * dL_dc_yy + u0 * dL_dc_xy;
        }
        /* W[i][j] (glm) = Rwc[i][j] = vm[j*4+i] */
        const float* vm = viewmatrix;
        float dL_dJ00 = vm[0 * 4 + 0] * dT0[0] + vm[1 * 4 + 0] * dT0[1] + vm[2 * 4 + 0] * dT0[2];
        float dL_dJ02 = vm[0 * 4 + 2] * dT0[0] + vm[1 * 4 + 2] * dT0[1] + vm[2 * 4 + 2] * dT0[2];
        float dL_dJ11 = vm[0 * 4 + 1] * dT1[0] + vm[1 * 4 + 1] * dT1[1] + vm[2 * 4 + 1] * dT1[2];
        float dL_dJ12 = vm[0 * 4 + 2] * dT1[0] + vm[1 * 4 + 2] * dT1[1] + vm[2 * 4 + 2] * dT1[2];
        float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        float dL_dtx = x_grad_mul * -focal_x * tz2 * dL_dJ02;
        float dL_dty = y_grad_mul * -focal_y * tz2 * dL_dJ12;
        float dL_dtz = -focal_x * tz2 * dL_dJ00 - focal_y * tz2 * dL_dJ11 + (2 * focal_x * t[0]) * tz3 * dL_dJ02 +
                       (2 * focal_y * t[1]) * tz3 * dL_dJ12;
        if (dL_dinvdepths) dL_dtz -= dL_dinvdepths[idx] / (t[2] * t[2]);
        /* transformVec4x3Transpose auxiliary.h:101-109; ASSIGN (backward.cu:324) */
        float* dm = dL_dmeans3D + 3 * (size_t)idx;
        dm[0] = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
        dm[1] = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
        dm[2] = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

        /* K10 backward.cu:422-447 */
        const float* proj = projmatrix;
        float m_hom[4];
        transformPoint4x4(mean, proj, m_hom);
        float m_w = 1.0f / (m_hom[3] + 0.0000001f);
        float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
        float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
        const float g2x = dL_dmeans2D[3 * idx], g2y = dL_dmeans2D[3 * idx + 1];
        dm[0] += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dm[1] += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dm[2] += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        if (shs) sh_backward(idx, D, M, means3D, campos, shs, c->clamped, dL_dcolors, dL_dmeans3D, dL_dsh);
        if (scales) {
            /* computeCov3D bwd, backward.cu:329-392 */
            const float* q = rotations + 4 * (size_t)idx;
            const float r = q[0], x = q[1], y = q[2], z = q[3];
            float Rq[3][3];
            quat_to_Rq(q, Rq);
            const float s[3] = {scale_modifier * scales[3 * idx], scale_modifier * scales[3 * idx + 1], scale_modifier * scales[3 * idx + 2]};
            float Mm[3][3];
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) Mm[k][a] = s[k] * Rq[a][k];
            const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            float dM[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) dM[a][b] = 2.0f * (Mm[a][0] * dS[0][b] + Mm[a][1] * dS[1][b] + Mm[a][2] * dS[2][b]);
            float* ds = dL_dscales + 3 * (size_t)idx;
            for (int k = 0; k < 3; k++) ds[k] = Rq[0][k] * dM[k][0] + Rq[1][k] * dM[k][1] + Rq[2][k] * dM[k][2];
            float G[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) G[a][b] = s[a] * dM[a][b];
            float* dq = dL_drotations + 4 * (size_t)idx;
            dq[0] = 2 * z * (G[0][1] - G[1][0]) + 2 * y * (G[2][0] - G[0][2]) + 2 * x * (G[1][2] - G[2][1]);
            dq[1] = 2 * y * (G[1][0] + G[0][1]) + 2 * z * (G[2][0] + G[0][2]) + 2 * r * (G[1][2] - G[2][1]) - 4 * x * (G[2][2] + G[1][1]);
            dq[2] = 2 * x * (G[1][0] + G[0][1]) + 2 * r * (G[2][0] - G[0][2]) + 2 * z * (G[1][2] + G[2][1]) - 4 * y * (G[2][2] + G[0][0]);
            dq[3] = 2 * r * (G[0][1] - G[1][0]) + 2 * x * (G[2][0] + G[0][2]) + 2 * y * (G[1][2] + G[2][1]) - 4 * z * (G[1][1] + G[0][0]);
        }
    }
}

int ora_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ora_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
