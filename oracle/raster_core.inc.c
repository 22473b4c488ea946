/*
 * ORACLE (test infrastructure, not product code) -- scalar CPU restatement of the
 * differentiable Gaussian-splat rasterizer that EavianWoo/SinGS calls through
 * `diff_gaussian_rasterization` (reference call sites:
 * sings/rec/renderer/gs_renderer_single.py:69-95, gs_renderer_multiple.py:95-121).
 *
 * PARITY UNPINNED: the rasterizer is an un-vendored, un-pinned third-party dependency
 * (install_all.sh:22 -> graphdeco-inria/diff-gaussian-rasterization, default branch
 * `main`); its source is absent from /root/reference and the reference holds no golden
 * vectors for it.  This file restates the PUBLISHED algorithm of that package (digest in
 * SURVEY.md Appendix A.1-A.5).  It is validated by analytic cases, fp64 finite
 * differences of its explicit backward, and an independent vectorised twin whose gradients come
 * from autograd (oracle/raster_torch.py, compared in tests/test_oracle_raster_twin.py).
 *
 * This file is included twice by raster_oracle.c with REAL = float / double.
 * All arithmetic is written operation-by-operation (compile with -ffp-contract=off) so
 * that the HIP kernels can reproduce the integer outcomes (radii, tile rects, depth key
 * bits) bit for bit.
 *
 * Conventions: viewmatrix / projmatrix are the flat memory of the
 * row-major torch [4,4] tensors the reference builds as the TRANSPOSE of the usual
 * column-vector matrices (sings/rec/datasets/Customdataset.py:105,132-133), i.e. column
 * major M.  Quaternions are (r,x,y,z) = (w,x,y,z) of sings/rec/utils/geometry/rotations.py:38-66.
 */

#define SG_BLOCK 16

static inline void FN(xf4x3)(const REAL *p, const REAL *m, REAL *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}

static inline void FN(xf4x4)(const REAL *p, const REAL *m, REAL *o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* rotation matrix (row-major R[3*i+j]) of the UN-normalised quaternion, App. A.1 step 3 */
static inline void FN(quat_to_R)(const REAL *q, REAL *R)
{
    REAL r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = (REAL)1 - (REAL)2 * (y * y + z * z);
    R[1] = (REAL)2 * (x * y - r * z);
    R[2] = (REAL)2 * (x * z + r * y);
    R[3] = (REAL)2 * (x * y + r * z);
    R[4] = (REAL)1 - (REAL)2 * (x * x + z * z);
    R[5] = (REAL)2 * (y * z - r * x);
    R[6] = (REAL)2 * (x * z - r * y);
    R[7] = (REAL)2 * (y * z + r * x);
    R[8] = (REAL)1 - (REAL)2 * (x * x + y * y);
}

/* Sigma = R S^2 R^T computed as M^T M with M = S R^T (M[k][i] = s_k R[i][k]); 6 unique */
static inline void FN(cov3d)(const REAL *scale, REAL mod, const REAL *q, REAL *c6)
{
    REAL R[9], M[9];
    FN(quat_to_R)(q, R);
    REAL s[3] = { mod * scale[0], mod * scale[1], mod * scale[2] };
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 3; i++)
            M[3 * k + i] = s[k] * R[3 * i + k];
    REAL S[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            S[3 * i + j] = M[0 + i] * M[0 + j] + M[3 + i] * M[3 + j] + M[6 + i] * M[6 + j];
    c6[0] = S[0]; c6[1] = S[1]; c6[2] = S[2]; c6[3] = S[4]; c6[4] = S[5]; c6[5] = S[8];
}

/* Mm = J * Rv, the 2x3 matrix with cov2D = Mm Sigma Mm^T.  Returns clamped t in tc. */
static inline void FN(proj_jac)(const REAL *t_in, REAL fx, REAL fy, REAL tanfovx, REAL tanfovy,
                                const REAL *view, REAL *Mm /*6*/, REAL *tc /*3*/, int *xin, int *yin)
{
    REAL limx = (REAL)1.3 * tanfovx, limy = (REAL)1.3 * tanfovy;
    REAL txtz = t_in[0] / t_in[2], tytz = t_in[1] / t_in[2];
    *xin = !(txtz < -limx || txtz > limx);
    *yin = !(tytz < -limy || tytz > limy);
    REAL cx = txtz < -limx ? -limx : txtz; cx = cx > limx ? limx : cx;
    REAL cy = tytz < -limy ? -limy : tytz; cy = cy > limy ? limy : cy;
    tc[0] = cx * t_in[2]; tc[1] = cy * t_in[2]; tc[2] = t_in[2];
    REAL j00 = fx / tc[2], j02 = -(fx * tc[0]) / (tc[2] * tc[2]);
    REAL j11 = fy / tc[2], j12 = -(fy * tc[1]) / (tc[2] * tc[2]);
    /* Rv[i][k] = view[i + 4k] */
    for (int k = 0; k < 3; k++) {
        Mm[k]     = j00 * view[0 + 4 * k] + j02 * view[2 + 4 * k];
        Mm[3 + k] = j11 * view[1 + 4 * k] + j12 * view[2 + 4 * k];
    }
}

static inline void FN(cov2d)(const REAL *Mm, const REAL *c6, REAL *abc)
{
    /* V = Sigma (symmetric), tmp = Mm * V (2x3), cov = tmp * Mm^T */
    REAL V[9] = { c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5] };
    REAL tmp[6];
    for (int a = 0; a < 2; a++)
        for (int j = 0; j < 3; j++)
            tmp[3 * a + j] = Mm[3 * a + 0] * V[0 + j] + Mm[3 * a + 1] * V[3 + j] + Mm[3 * a + 2] * V[6 + j];
    abc[0] = tmp[0] * Mm[0] + tmp[1] * Mm[1] + tmp[2] * Mm[2];
    abc[1] = tmp[0] * Mm[3] + tmp[1] * Mm[4] + tmp[2] * Mm[5];
    abc[2] = tmp[3] * Mm[3] + tmp[4] * Mm[4] + tmp[5] * Mm[5];
    abc[0] += (REAL)0.3;
    abc[2] += (REAL)0.3;
}

static const double SG_C0 = 0.28209479177387814, SG_C1 = 0.4886025119029199;
static const double SG_C2[5] = { 1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                 -1.0925484305920792, 0.5462742152960396 };
static const double SG_C3[7] = { -0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                 -0.5900435899266435 };

/* SH basis values b[0..(deg+1)^2) for unit direction d; polynomial of
 * sings/rec/utils/visualize/spherical_harmonics.py:87-113 */
static inline void FN(sh_basis)(int deg, const REAL *d, REAL *b)
{
    REAL x = d[0], y = d[1], z = d[2];
    b[0] = (REAL)SG_C0;
    if (deg > 0) {
        b[1] = -(REAL)SG_C1 * y; b[2] = (REAL)SG_C1 * z; b[3] = -(REAL)SG_C1 * x;
        if (deg > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = (REAL)SG_C2[0] * xy;
            b[5] = (REAL)SG_C2[1] * yz;
            b[6] = (REAL)SG_C2[2] * ((REAL)2 * zz - xx - yy);
            b[7] = (REAL)SG_C2[3] * xz;
            b[8] = (REAL)SG_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9]  = (REAL)SG_C3[0] * y * ((REAL)3 * xx - yy);
                b[10] = (REAL)SG_C3[1] * xy * z;
                b[11] = (REAL)SG_C3[2] * y * ((REAL)4 * zz - xx - yy);
                b[12] = (REAL)SG_C3[3] * z * ((REAL)2 * zz - (REAL)3 * xx - (REAL)3 * yy);
                b[13] = (REAL)SG_C3[4] * x * ((REAL)4 * zz - xx - yy);
                b[14] = (REAL)SG_C3[5] * z * (xx - yy);
                b[15] = (REAL)SG_C3[6] * x * (xx - (REAL)3 * yy);
            }
        }
    }
}

/* d(basis)/d(dir) : db[k][3] */
static inline void FN(sh_basis_grad)(int deg, const REAL *d, REAL *db /* 16*3 */)
{
    REAL x = d[0], y = d[1], z = d[2];
    for (int i = 0; i < 48; i++) db[i] = 0;
    if (deg > 0) {
        db[3 * 1 + 1] = -(REAL)SG_C1; db[3 * 2 + 2] = (REAL)SG_C1; db[3 * 3 + 0] = -(REAL)SG_C1;
        if (deg > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y;
            db[3 * 4 + 0] = (REAL)SG_C2[0] * y; db[3 * 4 + 1] = (REAL)SG_C2[0] * x;
            db[3 * 5 + 1] = (REAL)SG_C2[1] * z; db[3 * 5 + 2] = (REAL)SG_C2[1] * y;
            db[3 * 6 + 0] = (REAL)SG_C2[2] * (REAL)-2 * x; db[3 * 6 + 1] = (REAL)SG_C2[2] * (REAL)-2 * y;
            db[3 * 6 + 2] = (REAL)SG_C2[2] * (REAL)4 * z;
            db[3 * 7 + 0] = (REAL)SG_C2[3] * z; db[3 * 7 + 2] = (REAL)SG_C2[3] * x;
            db[3 * 8 + 0] = (REAL)SG_C2[4] * (REAL)2 * x; db[3 * 8 + 1] = (REAL)SG_C2[4] * (REAL)-2 * y;
            if (deg > 2) {
                db[3 * 9 + 0] = (REAL)SG_C3[0] * (REAL)6 * xy;
                db[3 * 9 + 1] = (REAL)SG_C3[0] * ((REAL)3 * xx - (REAL)3 * yy);
                db[3 * 10 + 0] = (REAL)SG_C3[1] * y * z; db[3 * 10 + 1] = (REAL)SG_C3[1] * x * z;
                db[3 * 10 + 2] = (REAL)SG_C3[1] * xy;
                db[3 * 11 + 0] = (REAL)SG_C3[2] * (REAL)-2 * xy;
                db[3 * 11 + 1] = (REAL)SG_C3[2] * ((REAL)4 * zz - xx - (REAL)3 * yy);
                db[3 * 11 + 2] = (REAL)SG_C3[2] * (REAL)8 * y * z;
                db[3 * 12 + 0] = (REAL)SG_C3[3] * (REAL)-6 * x * z;
                db[3 * 12 + 1] = (REAL)SG_C3[3] * (REAL)-6 * y * z;
                db[3 * 12 + 2] = (REAL)SG_C3[3] * ((REAL)6 * zz - (REAL)3 * xx - (REAL)3 * yy);
                db[3 * 13 + 0] = (REAL)SG_C3[4] * ((REAL)4 * zz - (REAL)3 * xx - yy);
                db[3 * 13 + 1] = (REAL)SG_C3[4] * (REAL)-2 * xy;
                db[3 * 13 + 2] = (REAL)SG_C3[4] * (REAL)8 * x * z;
                db[3 * 14 + 0] = (REAL)SG_C3[5] * (REAL)2 * x * z;
                db[3 * 14 + 1] = (REAL)SG_C3[5] * (REAL)-2 * y * z;
                db[3 * 14 + 2] = (REAL)SG_C3[5] * (xx - yy);
                db[3 * 15 + 0] = (REAL)SG_C3[6] * ((REAL)3 * xx - (REAL)3 * yy);
                db[3 * 15 + 1] = (REAL)SG_C3[6] * (REAL)-6 * xy;
            }
        }
    }
}

/* ------------------------------------------------------------------ A.1 forward preprocess */
void FN(sgo_preprocess)(int P, int D, int M,
                        const REAL *means3D, const REAL *scales, REAL scale_modifier,
                        const REAL *rotations, const REAL *opacities, const REAL *shs,
                        const REAL *colors_precomp, const REAL *cov3D_precomp,
                        const REAL *view, const REAL *proj, const REAL *campos,
                        int W, int H, REAL tanfovx, REAL tanfovy,
                        int32_t *radii, REAL *xy, REAL *depths, REAL *cov3D, REAL *rgb,
                        REAL *conic_opacity, uint8_t *clamped, uint32_t *tiles_touched,
                        int32_t *rect)
{
    const REAL fx = (REAL)W / ((REAL)2 * tanfovx), fy = (REAL)H / ((REAL)2 * tanfovy);
    const int gx = (W + SG_BLOCK - 1) / SG_BLOCK, gy = (H + SG_BLOCK - 1) / SG_BLOCK;
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0;
        if (rect) { rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0; }
        const REAL *p = means3D + 3 * i;
        REAL pv[3];
        FN(xf4x3)(p, view, pv);
        if (pv[2] <= (REAL)0.2) continue;                       /* near cull (step 1) */
        REAL ph[4];
        FN(xf4x4)(p, proj, ph);
        REAL pw = (REAL)1 / (ph[3] + (REAL)0.0000001);
        REAL pp[3] = { ph[0] * pw, ph[1] * pw, ph[2] * pw };
        REAL c6[6];
        if (cov3D_precomp) { for (int k = 0; k < 6; k++) c6[k] = cov3D_precomp[6 * i + k]; }
        else { FN(cov3d)(scales + 3 * i, scale_modifier, rotations + 4 * i, c6); }
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = c6[k];
        REAL Mm[6], tc[3], abc[3]; int xin, yin;
        FN(proj_jac)(pv, fx, fy, tanfovx, tanfovy, view, Mm, tc, &xin, &yin);
        FN(cov2d)(Mm, c6, abc);
        REAL det = abc[0] * abc[2] - abc[1] * abc[1];
        if (det == (REAL)0) continue;
        REAL det_inv = (REAL)1 / det;
        REAL conic[3] = { abc[2] * det_inv, -abc[1] * det_inv, abc[0] * det_inv };
        REAL mid = (REAL)0.5 * (abc[0] + abc[2]);
        REAL dd = mid * mid - det; if (dd < (REAL)0.1) dd = (REAL)0.1;
        REAL sq = SQRT(dd);
        REAL l1 = mid + sq, l2 = mid - sq;
        REAL lm = l1 > l2 ? l1 : l2;
        REAL my_radius = CEIL((REAL)3 * SQRT(lm));
        /* non-finite / non-positive / unrepresentable radius (NaN or Inf covariance): culled.  Upstream's (int)NaN is 0 on its
         * hardware: radii = 0, no keys emitted -- nothing rendered; (int) of such a value is undefined in C, so it is decided here */
        if (!(my_radius > (REAL)0 && my_radius < (REAL)1073741824.0)) continue;
        REAL pix[2] = { ((pp[0] + (REAL)1) * (REAL)W - (REAL)1) * (REAL)0.5,
                        ((pp[1] + (REAL)1) * (REAL)H - (REAL)1) * (REAL)0.5 };
        int mr = (int)my_radius;
        int rminx = (int)((pix[0] - (REAL)mr) / (REAL)SG_BLOCK);
        int rminy = (int)((pix[1] - (REAL)mr) / (REAL)SG_BLOCK);
        int rmaxx = (int)((pix[0] + (REAL)mr + (REAL)(SG_BLOCK - 1)) / (REAL)SG_BLOCK);
        int rmaxy = (int)((pix[1] + (REAL)mr + (REAL)(SG_BLOCK - 1)) / (REAL)SG_BLOCK);
        rminx = rminx < 0 ? 0 : rminx; rminx = rminx > gx ? gx : rminx;
        rminy = rminy < 0 ? 0 : rminy; rminy = rminy > gy ? gy : rminy;
        rmaxx = rmaxx < 0 ? 0 : rmaxx; rmaxx = rmaxx > gx ? gx : rmaxx;
        rmaxy = rmaxy < 0 ? 0 : rmaxy; rmaxy = rmaxy > gy ? gy : rmaxy;
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        if (colors_precomp) {
            for (int c = 0; c < 3; c++) { rgb[3 * i + c] = colors_precomp[3 * i + c]; clamped[3 * i + c] = 0; }
        } else {
            REAL dir[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
            REAL len = SQRT(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
            REAL b[16];
            FN(sh_basis)(D, dir, b);
            int nc = (D + 1) * (D + 1);
            const REAL *sh = shs + (size_t)i * M * 3;
            for (int c = 0; c < 3; c++) {
                REAL acc = b[0] * sh[c];
                for (int k = 1; k < nc; k++) acc = acc + b[k] * sh[3 * k + c];
                acc = acc + (REAL)0.5;
                clamped[3 * i + c] = acc < (REAL)0;
                rgb[3 * i + c] = acc < (REAL)0 ? (REAL)0 : acc;
            }
        }
        depths[i] = pv[2];
        radii[i] = mr;
        xy[2 * i] = pix[0]; xy[2 * i + 1] = pix[1];
        conic_opacity[4 * i + 0] = conic[0]; conic_opacity[4 * i + 1] = conic[1];
        conic_opacity[4 * i + 2] = conic[2]; conic_opacity[4 * i + 3] = opacities[i];
        tiles_touched[i] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
        if (rect) { rect[4 * i] = rminx; rect[4 * i + 1] = rminy; rect[4 * i + 2] = rmaxx; rect[4 * i + 3] = rmaxy; }
    }
}

/* ------------------------------------------------------------------ A.3 forward render */
/* margin[pix] (optional): smallest relative distance of any threshold decision taken for
 * this pixel (alpha vs 1/255, test_T vs 1e-4, power vs 0) -- lets tests tell a genuine
 * mismatch from a borderline flip caused by a 1-ulp exp difference.
 * flip[pix] (optional, with `border`): an upper bound on how far the pixel's colour can move if
 * the decisions whose margin is below `border` are taken the other way -- per such decision the
 * contribution of the one splat concerned, alpha T (|c| + cmax) (cmax = largest colour / background
 * component: what can stand behind it), or 2 T cmax for the early-termination test (the tail
 * behind it is blended or replaced by the background). */
void FN(sgo_render_fwd)(int W, int H, const uint32_t *ranges, const uint32_t *point_list,
                        const REAL *xy, const REAL *feat, const REAL *conic_opacity,
                        const REAL *bg, REAL *out_color, REAL *final_T, uint32_t *n_contrib,
                        REAL *margin, REAL border, REAL cmax, REAL *flip)
{
    const int gx = (W + SG_BLOCK - 1) / SG_BLOCK;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            int tile = (py / SG_BLOCK) * gx + (px / SG_BLOCK);
            uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
            REAL T = 1, C[3] = { 0, 0, 0 }, mg = 1, fl = 0;
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = s; k < e; k++) {
                contributor++;
                uint32_t g = point_list[k];
                REAL dx = xy[2 * g] - (REAL)px, dy = xy[2 * g + 1] - (REAL)py;
                const REAL *co = conic_opacity + 4 * g;
                REAL q1 = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy), q2 = co[1] * dx * dy;
                REAL power = q1 - q2;
                REAL pscale = FABS(q1) + FABS(q2);
                REAL cabs = FABS(feat[3 * g]);
                if (FABS(feat[3 * g + 1]) > cabs) cabs = FABS(feat[3 * g + 1]);
                if (FABS(feat[3 * g + 2]) > cabs) cabs = FABS(feat[3 * g + 2]);
                if (pscale > 0) {
                    REAL m = FABS(power) / pscale;
                    if (power > (REAL)-1e-3) {
                        if (m < mg) mg = m;
                        if (m < border) { REAL a0 = co[3] > (REAL)0.99 ? (REAL)0.99 : co[3]; fl += a0 * T * (cabs + cmax); }
                    }
                }
                if (power > 0) continue;
                REAL alpha = co[3] * EXP(power);
                if (alpha > (REAL)0.99) alpha = (REAL)0.99;
                { REAL m = FABS(alpha * (REAL)255 - (REAL)1); if (m < mg) mg = m; if (m < border) fl += alpha * T * (cabs + cmax); }
                if (alpha < (REAL)1 / (REAL)255) continue;
                REAL test_T = T * ((REAL)1 - alpha);
                { REAL m = FABS(test_T * (REAL)10000 - (REAL)1); if (m < mg) mg = m; if (m < border) fl += (REAL)2 * T * cmax; }
                if (test_T < (REAL)0.0001) break;
                for (int c = 0; c < 3; c++) C[c] += feat[3 * g + c] * alpha * T;
                T = test_T;
                last = contributor;
            }
            size_t pid = (size_t)py * W + px;
            final_T[pid] = T; n_contrib[pid] = last;
            for (int c = 0; c < 3; c++) out_color[(size_t)c * H * W + pid] = C[c] + T * bg[c];
            if (margin) margin[pid] = mg;
            if (flip) flip[pid] = fl;
        }
}

/* ------------------------------------------------------------------ A.4 backward render */
/* Sums go through double accumulators in a fixed order (upstream uses order-dependent
 * float atomics). dL_dconic has 4 slots per Gaussian (.z unused), as upstream. */
void FN(sgo_render_bwd)(int P, int W, int H, const uint32_t *ranges, const uint32_t *point_list,
                        const REAL *xy, const REAL *conic_opacity, const REAL *colors,
                        const REAL *bg, const REAL *final_T, const uint32_t *n_contrib,
                        const REAL *dL_dpix,
                        REAL *dL_dmean2D /*P*3*/, REAL *dL_dconic /*P*4*/,
                        REAL *dL_dopacity /*P*/, REAL *dL_dcolor /*P*3*/)
{
    const int gx = (W + SG_BLOCK - 1) / SG_BLOCK;
    double *acc = (double *)calloc((size_t)P * 9, sizeof(double));
    const REAL ddelx_dx = (REAL)0.5 * (REAL)W, ddely_dy = (REAL)0.5 * (REAL)H;
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            int tile = (py / SG_BLOCK) * gx + (px / SG_BLOCK);
            uint32_t s = ranges[2 * tile], e = ranges[2 * tile + 1];
            size_t pid = (size_t)py * W + px;
            REAL T_final = final_T[pid], T = T_final;
            uint32_t last_contributor = n_contrib[pid];
            REAL accum_rec[3] = { 0, 0, 0 }, last_color[3] = { 0, 0, 0 }, last_alpha = 0;
            REAL dLp[3] = { dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[(size_t)2 * H * W + pid] };
            REAL bg_dot = bg[0] * dLp[0] + bg[1] * dLp[1] + bg[2] * dLp[2];
            uint32_t contributor = e - s;
            for (uint32_t k = e; k-- > s;) {
                contributor--;
                if (contributor >= last_contributor) continue;
                uint32_t g = point_list[k];
                REAL dx = xy[2 * g] - (REAL)px, dy = xy[2 * g + 1] - (REAL)py;
                const REAL *co = conic_opacity + 4 * g;
                REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0) continue;
                REAL G = EXP(power);
                REAL alpha = co[3] * G; if (alpha > (REAL)0.99) alpha = (REAL)0.99;
                if (alpha < (REAL)1 / (REAL)255) continue;
                T = T / ((REAL)1 - alpha);
                REAL dchannel_dcolor = alpha * T;
                REAL dL_dalpha = 0;
                for (int c = 0; c < 3; c++) {
                    REAL col = colors[3 * g + c];
                    accum_rec[c] = last_alpha * last_color[c] + ((REAL)1 - last_alpha) * accum_rec[c];
                    last_color[c] = col;
                    dL_dalpha += (col - accum_rec[c]) * dLp[c];
                    acc[(size_t)g * 9 + 6 + c] += (double)(dchannel_dcolor * dLp[c]);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / ((REAL)1 - alpha)) * bg_dot;
                REAL dL_dG = co[3] * dL_dalpha;
                REAL gdx = G * dx, gdy = G * dy;
                REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                acc[(size_t)g * 9 + 0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                acc[(size_t)g * 9 + 1] += (double)(dL_dG * dG_ddely * ddely_dy);
                acc[(size_t)g * 9 + 2] += (double)((REAL)-0.5 * gdx * dx * dL_dG);
                acc[(size_t)g * 9 + 3] += (double)((REAL)-0.5 * gdx * dy * dL_dG);
                acc[(size_t)g * 9 + 4] += (double)((REAL)-0.5 * gdy * dy * dL_dG);
                acc[(size_t)g * 9 + 5] += (double)(G * dL_dalpha);
            }
        }
    for (int g = 0; g < P; g++) {
        const double *a = acc + (size_t)g * 9;
        dL_dmean2D[3 * g] = (REAL)a[0]; dL_dmean2D[3 * g + 1] = (REAL)a[1]; dL_dmean2D[3 * g + 2] = 0;
        dL_dconic[4 * g] = (REAL)a[2]; dL_dconic[4 * g + 1] = (REAL)a[3];
        dL_dconic[4 * g + 2] = 0; dL_dconic[4 * g + 3] = (REAL)a[4];
        dL_dopacity[g] = (REAL)a[5];
        dL_dcolor[3 * g] = (REAL)a[6]; dL_dcolor[3 * g + 1] = (REAL)a[7]; dL_dcolor[3 * g + 2] = (REAL)a[8];
    }
    free(acc);
}

/* ------------------------------------------------------------------ A.5 backward preprocess */
/* dL_dconic: [P,4] (x, y, unused, w);  dL_dcolor: [P,3];  dL_dmean2D: [P,3].
 * Outputs are overwritten for every Gaussian (zero where radii == 0). */
void FN(sgo_preprocess_bwd)(int P, int D, int M,
                            const REAL *means3D, const int32_t *radii, const REAL *shs,
                            const uint8_t *clamped, const REAL *scales, const REAL *rotations,
                            REAL scale_modifier, const REAL *cov3D, int cov3D_is_precomp,
                            int colors_are_precomp,
                            const REAL *view, const REAL *proj, const REAL *campos,
                            int W, int H, REAL tanfovx, REAL tanfovy,
                            const REAL *dL_dmean2D, const REAL *dL_dconic, const REAL *dL_dcolor,
                            REAL *dL_dmeans3D, REAL *dL_dcov3D, REAL *dL_dsh,
                            REAL *dL_dscales, REAL *dL_drots)
{
    const REAL fx = (REAL)W / ((REAL)2 * tanfovx), fy = (REAL)H / ((REAL)2 * tanfovy);
    for (int i = 0; i < P; i++) {
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = 0;
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0;
        if (dL_dsh) for (int k = 0; k < M * 3; k++) dL_dsh[(size_t)i * M * 3 + k] = 0;
        if (dL_dscales) for (int k = 0; k < 3; k++) dL_dscales[3 * i + k] = 0;
        if (dL_drots) for (int k = 0; k < 4; k++) dL_drots[4 * i + k] = 0;
        if (!(radii[i] > 0)) continue;
        const REAL *p = means3D + 3 * i;
        const REAL *c6 = cov3D + 6 * i;
        /* ---- cov2D backward (computeCov2D) */
        REAL pv[3], Mm[6], tc[3], abc[3]; int xin, yin;
        FN(xf4x3)(p, view, pv);
        FN(proj_jac)(pv, fx, fy, tanfovx, tanfovy, view, Mm, tc, &xin, &yin);
        FN(cov2d)(Mm, c6, abc);
        REAL a = abc[0], b = abc[1], c = abc[2];
        REAL dLx = dL_dconic[4 * i], dLy = dL_dconic[4 * i + 1], dLz = dL_dconic[4 * i + 3];
        REAL denom = a * c - b * b;
        REAL denom2inv = (REAL)1 / (denom * denom + (REAL)0.0000001);
        REAL dL_da = 0, dL_db = 0, dL_dc = 0;
        REAL gM[6] = { 0, 0, 0, 0, 0, 0 };     /* dL/dMm */
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dLx + (REAL)2 * b * c * dLy + (denom - a * c) * dLz);
            dL_dc = denom2inv * (-a * a * dLz + (REAL)2 * a * b * dLy + (denom - a * c) * dLx);
            dL_db = denom2inv * (REAL)2 * (b * c * dLx - (denom + (REAL)2 * b * b) * dLy + a * b * dLz);
            const REAL *m0 = Mm, *m1 = Mm + 3;
            REAL *g6 = dL_dcov3D + 6 * i;
            g6[0] = m0[0] * m0[0] * dL_da + m0[0] * m1[0] * dL_db + m1[0] * m1[0] * dL_dc;
            g6[3] = m0[1] * m0[1] * dL_da + m0[1] * m1[1] * dL_db + m1[1] * m1[1] * dL_dc;
            g6[5] = m0[2] * m0[2] * dL_da + m0[2] * m1[2] * dL_db + m1[2] * m1[2] * dL_dc;
            g6[1] = (REAL)2 * m0[0] * m0[1] * dL_da + (m0[0] * m1[1] + m0[1] * m1[0]) * dL_db + (REAL)2 * m1[0] * m1[1] * dL_dc;
            g6[2] = (REAL)2 * m0[0] * m0[2] * dL_da + (m0[0] * m1[2] + m0[2] * m1[0]) * dL_db + (REAL)2 * m1[0] * m1[2] * dL_dc;
            g6[4] = (REAL)2 * m0[2] * m0[1] * dL_da + (m0[1] * m1[2] + m0[2] * m1[1]) * dL_db + (REAL)2 * m1[1] * m1[2] * dL_dc;
        }
        {
            REAL V[9] = { c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5] };
            for (int k = 0; k < 3; k++) {
                REAL v0 = Mm[0] * V[3 * k] + Mm[1] * V[3 * k + 1] + Mm[2] * V[3 * k + 2];   /* (V Mm0)_k */
                REAL v1 = Mm[3] * V[3 * k] + Mm[4] * V[3 * k + 1] + Mm[5] * V[3 * k + 2];
                gM[k]     = (REAL)2 * v0 * dL_da + v1 * dL_db;
                gM[3 + k] = (REAL)2 * v1 * dL_dc + v0 * dL_db;
            }
        }
        /* Mm = J Rv: dL/dJ[a][b] = sum_k gM[a][k] Rv[b][k], Rv[b][k] = view[b + 4k] */
        REAL dJ00 = gM[0] * view[0] + gM[1] * view[4] + gM[2] * view[8];
        REAL dJ02 = gM[0] * view[2] + gM[1] * view[6] + gM[2] * view[10];
        REAL dJ11 = gM[3] * view[1] + gM[4] * view[5] + gM[5] * view[9];
        REAL dJ12 = gM[3] * view[2] + gM[4] * view[6] + gM[5] * view[10];
        REAL tz = (REAL)1 / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
        REAL dtx = (REAL)xin * -fx * tz2 * dJ02;
        REAL dty = (REAL)yin * -fy * tz2 * dJ12;
        REAL dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + ((REAL)2 * fx * tc[0]) * tz3 * dJ02 + ((REAL)2 * fy * tc[1]) * tz3 * dJ12;
        /* transformVec4x3Transpose */
        REAL dmean[3] = { view[0] * dtx + view[1] * dty + view[2] * dtz,
                          view[4] * dtx + view[5] * dty + view[6] * dtz,
                          view[8] * dtx + view[9] * dty + view[10] * dtz };
        /* ---- projection backward */
        REAL mh[4];
        FN(xf4x4)(p, proj, mh);
        REAL mw = (REAL)1 / (mh[3] + (REAL)0.0000001);
        REAL mul1 = (proj[0] * p[0] + proj[4] * p[1] + proj[8] * p[2] + proj[12]) * mw * mw;
        REAL mul2 = (proj[1] * p[0] + proj[5] * p[1] + proj[9] * p[2] + proj[13]) * mw * mw;
        REAL g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
        dmean[0] += (proj[0] * mw - proj[3] * mul1) * g2x + (proj[1] * mw - proj[3] * mul2) * g2y;
        dmean[1] += (proj[4] * mw - proj[7] * mul1) * g2x + (proj[5] * mw - proj[7] * mul2) * g2y;
        dmean[2] += (proj[8] * mw - proj[11] * mul1) * g2x + (proj[9] * mw - proj[11] * mul2) * g2y;
        /* ---- SH backward */
        if (!colors_are_precomp) {
            REAL dorig[3] = { p[0] - campos[0], p[1] - campos[1], p[2] - campos[2] };
            REAL len = SQRT(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
            REAL dir[3] = { dorig[0] / len, dorig[1] / len, dorig[2] / len };
            REAL bas[16], db[48];
            FN(sh_basis)(D, dir, bas);
            FN(sh_basis_grad)(D, dir, db);
            int nc = (D + 1) * (D + 1);
            const REAL *sh = shs + (size_t)i * M * 3;
            REAL dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * i + ch] ? (REAL)0 : dL_dcolor[3 * i + ch];
            REAL ddir[3] = { 0, 0, 0 };
            for (int k = 0; k < nc; k++)
                for (int ch = 0; ch < 3; ch++) {
                    dL_dsh[(size_t)i * M * 3 + 3 * k + ch] = bas[k] * dRGB[ch];
                    for (int ax = 0; ax < 3; ax++) ddir[ax] += db[3 * k + ax] * sh[3 * k + ch] * dRGB[ch];
                }
            /* dnormvdv */
            REAL sum2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
            REAL inv32 = (REAL)1 / SQRT(sum2 * sum2 * sum2);
            REAL vx = dorig[0], vy = dorig[1], vz = dorig[2];
            dmean[0] += ((sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * inv32;
            dmean[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * inv32;
            dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * inv32;
        }
        for (int k = 0; k < 3; k++) dL_dmeans3D[3 * i + k] = dmean[k];
        /* ---- cov3D backward: Sigma = R S^2 R^T */
        if (!cov3D_is_precomp) {
            const REAL *g6 = dL_dcov3D + 6 * i;
            REAL Gs[9] = { g6[0], (REAL)0.5 * g6[1], (REAL)0.5 * g6[2],
                           (REAL)0.5 * g6[1], g6[3], (REAL)0.5 * g6[4],
                           (REAL)0.5 * g6[2], (REAL)0.5 * g6[4], g6[5] };
            REAL R[9];
            const REAL *q = rotations + 4 * i;
            FN(quat_to_R)(q, R);
            REAL s[3] = { scale_modifier * scales[3 * i], scale_modifier * scales[3 * i + 1], scale_modifier * scales[3 * i + 2] };
            /* GR = Gs R ; dL/ds_k = 2 s_k (R^T Gs R)_kk ; dL/dR = 2 Gs R S^2 */
            REAL GR[9], dR[9];
            for (int a2 = 0; a2 < 3; a2++)
                for (int k = 0; k < 3; k++)
                    GR[3 * a2 + k] = Gs[3 * a2] * R[k] + Gs[3 * a2 + 1] * R[3 + k] + Gs[3 * a2 + 2] * R[6 + k];
            for (int k = 0; k < 3; k++) {
                REAL rgr = R[k] * GR[k] + R[3 + k] * GR[3 + k] + R[6 + k] * GR[6 + k];
                /* upstream's computeCov3D backward: dot(Rt[k], dL_dMt[k]) = the gradient w.r.t. the MODIFIED scale
                 * s = mod * scale, reported as dL/dscale without the factor mod (the derivative of its own forward would
                 * carry it; identical at mod = 1, the only value the reference differentiates at) */
                dL_dscales[3 * i + k] = (REAL)2 * s[k] * rgr;
                for (int a2 = 0; a2 < 3; a2++) dR[3 * a2 + k] = (REAL)2 * GR[3 * a2 + k] * s[k] * s[k];
            }
            REAL r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drots[4 * i + 0] = (REAL)2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            dL_drots[4 * i + 1] = (REAL)2 * (y * dR[1] + z * dR[2] + y * dR[3] - (REAL)2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - (REAL)2 * x * dR[8]);
            dL_drots[4 * i + 2] = (REAL)2 * ((REAL)-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - (REAL)2 * y * dR[8]);
            dL_drots[4 * i + 3] = (REAL)2 * ((REAL)-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - (REAL)2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
}
