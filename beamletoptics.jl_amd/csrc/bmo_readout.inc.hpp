// bmo_readout.inc.hpp — detector read-out kernels (included by bmo_engine.hip; shares its pool / error helpers).
//
// PSF intensity (PSFDetector.jl:190-237): an n x n grid of sample points times H recorded hits, one cis() per pair.
//   * a workgroup owns 256 consecutive grid points (point index = i + n*j, i fastest like the reference's Matrix) and a
//     contiguous range of hits ("split"); hits are staged through LDS in tiles of 256 x 9 doubles so every lane reads each
//     hit from LDS (broadcast reads) instead of HBM;
//   * each lane accumulates its point's complex sum over the split in hit order; the per-split partial sums are written to
//     a [split][point] plane and reduced in split order by a second kernel, which also takes abs2.  The result is
//     deterministic for a given (n, H); it re-associates the reference's sequential sum at split boundaries only.
//   * arithmetic follows the reference expression by expression (p = origin + x*e1 + z*e2; l = dot(p - hit, dir);
//     phase = k*(opl + l); acc += proj*cis(phase)), FP64, no contraction.
namespace {

constexpr int PSF_TILE = 256;

__global__ __launch_bounds__(256) void psf_accumulate_kernel(const double* __restrict__ hits, int64_t n_hits, int64_t hits_per_split, const double* __restrict__ xs,
                                                             const double* __restrict__ zs, int32_t n, d3 origin, d3 e1, d3 e2, double2* __restrict__ partial) {
    __shared__ double tile[PSF_TILE * 9];
    const int64_t n_pts = (int64_t)n * n;
    const int64_t pt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = pt < n_pts;
    double px = 0, py = 0, pz = 0;
    if (live) {
        const int i = (int)(pt % n), j = (int)(pt / n);
        const double x = xs[i], z = zs[j];
        // origin_pd + x * e1 + z * e2  (left to right)
        px = (origin.x + x * e1.x) + z * e2.x;
        py = (origin.y + x * e1.y) + z * e2.y;
        pz = (origin.z + x * e1.z) + z * e2.z;
    }
    const int64_t h0 = (int64_t)blockIdx.y * hits_per_split;
    const int64_t h1 = h0 + hits_per_split < n_hits ? h0 + hits_per_split : n_hits;
    double re = 0.0, im = 0.0;
    for (int64_t base = h0; base < h1; base += PSF_TILE) {
        const int cnt = (int)(h1 - base < PSF_TILE ? h1 - base : PSF_TILE);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt * 9; q += 256) tile[q] = hits[base * 9 + q];
        __syncthreads();
        if (live) {
            for (int h = 0; h < cnt; ++h) {
                const double* r = tile + 9 * h;
                const double l = ((px - r[0]) * r[3] + (py - r[1]) * r[4]) + (pz - r[2]) * r[5];
                const double phase = r[8] * (r[6] + l);
                double s, c;
                sincos(phase, &s, &c);
                re += r[7] * c;
                im += r[7] * s;
            }
        }
    }
    if (live) partial[(int64_t)blockIdx.y * n_pts + pt] = make_double2(re, im);
}

__global__ void psf_reduce_kernel(const double2* __restrict__ partial, int32_t n_splits, int64_t n_pts, double* __restrict__ intensity, double2* __restrict__ field) {
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n_pts) return;
    double re = 0.0, im = 0.0;
    for (int s = 0; s < n_splits; ++s) {
        const double2 v = partial[(int64_t)s * n_pts + pt];
        re += v.x;
        im += v.y;
    }
    intensity[pt] = re * re + im * im;  // abs2
    if (field) field[pt] = make_double2(re, im);
}

}  // namespace

extern "C" int bmo_psf_intensity(const double* hits, int64_t n_hits, int32_t hits_on_device, const double origin[3], const double e1[3], const double e2[3],
                                 const double* xs, const double* zs, int32_t n, int32_t device, double* out_intensity, double* out_field, double* kernel_ms) {
    if (!origin || !e1 || !e2 || !xs || !zs || !out_intensity || n <= 0 || n_hits < 0 || (n_hits > 0 && !hits))
        return fail(BMO_ERR_INVALID, "bmo_psf_intensity: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BMO_ERR_NO_DEVICE, "bmo_psf_intensity: no HIP device (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(BMO_ERR_INVALID, "bmo_psf_intensity: bad device ordinal");
    HIP_TRY(hipSetDevice(device));
    const int64_t n_pts = (int64_t)n * n;
    const unsigned pt_blocks = (unsigned)((n_pts + 255) / 256);
    // enough workgroups to fill 256 CUs several times over, but never splits shorter than one LDS tile
    int64_t n_splits = (4096 + pt_blocks - 1) / pt_blocks;
    const int64_t max_splits = (n_hits + PSF_TILE - 1) / PSF_TILE;
    if (n_splits > max_splits) n_splits = max_splits;
    if (n_splits < 1) n_splits = 1;
    if (n_splits > 65535) n_splits = 65535;
    int64_t hits_per_split = (n_hits + n_splits - 1) / n_splits;
    hits_per_split = (hits_per_split + PSF_TILE - 1) / PSF_TILE * PSF_TILE;
    if (hits_per_split < PSF_TILE) hits_per_split = PSF_TILE;
    n_splits = n_hits > 0 ? (n_hits + hits_per_split - 1) / hits_per_split : 1;

    DevBuf d_hits, d_xs, d_zs, d_partial, d_int, d_field;
    const double* hits_dev = hits;
    int rc;
    if (!hits_on_device && n_hits > 0) {
        if ((rc = d_hits.alloc((size_t)n_hits * 9 * sizeof(double)))) return rc;
        HIP_TRY(hipMemcpy(d_hits.p, hits, (size_t)n_hits * 9 * sizeof(double), hipMemcpyHostToDevice));
        hits_dev = (const double*)d_hits.p;
    }
    if ((rc = d_xs.alloc((size_t)n * sizeof(double))) || (rc = d_zs.alloc((size_t)n * sizeof(double))) ||
        (rc = d_partial.alloc((size_t)n_splits * n_pts * sizeof(double2))) || (rc = d_int.alloc((size_t)n_pts * sizeof(double))))
        return rc;
    if (out_field && (rc = d_field.alloc((size_t)n_pts * sizeof(double2)))) return rc;
    HIP_TRY(hipMemcpy(d_xs.p, xs, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_zs.p, zs, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipEvent_t e0, e1v;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1v));
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(psf_accumulate_kernel, dim3(pt_blocks, (unsigned)n_splits), dim3(256), 0, 0, hits_dev, n_hits, hits_per_split, (const double*)d_xs.p,
                       (const double*)d_zs.p, n, d3{origin[0], origin[1], origin[2]}, d3{e1[0], e1[1], e1[2]}, d3{e2[0], e2[1], e2[2]}, (double2*)d_partial.p);
    hipLaunchKernelGGL(psf_reduce_kernel, dim3(pt_blocks), dim3(256), 0, 0, (const double2*)d_partial.p, (int32_t)n_splits, n_pts, (double*)d_int.p,
                       out_field ? (double2*)d_field.p : nullptr);
    HIP_TRY(hipEventRecord(e1v, 0));
    HIP_TRY(hipEventSynchronize(e1v));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1v));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1v);
    if (kernel_ms) *kernel_ms = ms;
    HIP_TRY(hipMemcpy(out_intensity, d_int.p, (size_t)n_pts * sizeof(double), hipMemcpyDeviceToHost));
    if (out_field) HIP_TRY(hipMemcpy(out_field, d_field.p, (size_t)n_pts * sizeof(double2), hipMemcpyDeviceToHost));
    return BMO_OK;
}

// ====================================================================================================================
// Photodetector field (Photodetector.jl:69-107).  Pipeline, all on the device the trace ran on:
//   1. pd_hit_nodes_kernel   : recorded beamlets of the slot (reference order) -> device node id, segment count, node->hit map
//   2. exclusive scan        : segment offsets of the per-hit segment table
//   3. pd_gather_kernel      : one pass over the step chunks of the segment log copies the chief / waist / divergence rays of
//                              those beamlets into a compact SoA table [24][total_segs] (+ the OPL carried in from the parent)
//   4. pd_prepare_kernel     : per beamlet, the hit-constant scalars in the reference's summation order: cumulative chief
//                              lengths (point_on_beam's `temp`), length(gauss), optical_path_length(gauss), l0, k, ref_phi
//   5. pd_field_kernel       : 256 grid points per workgroup x a contiguous range of beamlets; per pair the reference's
//                              expression sequence (point on the detector, projection on the beamlet axis, point_on_beam,
//                              gauss_parameters, electric_field); partial sums per beamlet range
//   6. pd_reduce_kernel      : sums the ranges in order and adds the result to the caller's field
namespace {

enum { PD_SEG_PLANES = 24, PD_HS = 12 };
// per-hit scalars: 0 l_parent 1 w0 2 E0.re 3 E0.im 4 lambda 5 proj 6 opl_parent | prepared: 7 l0 8 k 9 ref_phi 10 len_total 11 unused

// Earlier segments of continued root beamlets (bmo_result_set_gauss_prefix): root node nd owns rows pre_start[nd] .. pre_start[nd + 1] of the
// prefix table; they come in front of the segments of the log.  nullptr: no beamlet has any.
struct PdPrefix {
    const int32_t* start;  // [n_roots + 1]
    const double* segs;    // [24][total]
    const double* opl;     // [n_roots]: optical path length of the parent chain
    int64_t n_roots, total;
    __device__ int len(int32_t nd) const { return (start && nd < n_roots) ? start[nd + 1] - start[nd] : 0; }
};

__global__ void pd_hit_nodes_kernel(const int32_t* __restrict__ det_node, int64_t first_row, int64_t n_hits, const int32_t* __restrict__ order,
                                    const int32_t* __restrict__ nseg, const double* __restrict__ aux, const double* __restrict__ lambda,
                                    const double* __restrict__ det_data, int32_t* __restrict__ hit_node, int32_t* __restrict__ hit_nseg,
                                    int32_t* __restrict__ node_hit, double* __restrict__ hs, PdPrefix pre) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hits) return;
    const int64_t row = first_row + 3 * h;  // three hit rows per beamlet, row 0 = {proj, 0, ...}
    const int32_t nd = order[det_node[row]];
    hit_node[h] = nd;
    hit_nseg[h] = nseg[nd] + pre.len(nd);
    node_hit[nd] = (int32_t)h;
    double* s = hs + h * PD_HS;
    s[0] = aux[(int64_t)nd * 4 + 0];
    s[1] = aux[(int64_t)nd * 4 + 1];
    s[2] = aux[(int64_t)nd * 4 + 2];
    s[3] = aux[(int64_t)nd * 4 + 3];
    s[4] = lambda[nd];
    s[5] = det_data[row * 9 + 0];
}

__global__ void pd_gather_kernel(Chunk c, const int32_t* __restrict__ node_hit, const int32_t* __restrict__ seg_start, int64_t total_segs,
                                 double* __restrict__ segs, double* __restrict__ hs, PdPrefix pre) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.count) return;
    const int32_t nd = chunk_node(c, j);
    if (nd < 0) return;  // hole of a fused level
    const int32_t h = node_hit[nd];
    if (h < 0) return;
    const int32_t k = c.i[I_K * c.cap + j];
    const int plen = pre.len(nd);
    const int64_t dst = (int64_t)seg_start[h] + plen + k;
    for (int b = 0; b < 3; ++b)          // chief, waist, divergence: record planes 11*b + {pos 0-2, dir 3-5, n 6, t 7}
        for (int q = 0; q < 8; ++q) segs[(int64_t)(8 * b + q) * total_segs + dst] = c.d[(int64_t)(11 * b + q) * c.cap + j];
    if (k == 0) {
        // OPL of the parent chain (optical_path_length(parent)); a continued beamlet's record carries the OPL up to its open ray, the
        // parent's share of it came with the prefix
        hs[(int64_t)h * PD_HS + 6] = plen > 0 ? pre.opl[nd] : c.d[(int64_t)35 * c.cap + j];
        for (int i = 0; i < plen; ++i)
            for (int q = 0; q < PD_SEG_PLANES; ++q) segs[(int64_t)q * total_segs + seg_start[h] + i] = pre.segs[(int64_t)q * pre.total + pre.start[nd] + i];
    }
}

__global__ void pd_prepare_kernel(int64_t n_hits, const int32_t* __restrict__ seg_start, const int32_t* __restrict__ hit_nseg, int64_t total_segs,
                                  const double* __restrict__ segs, double* __restrict__ cum, double* __restrict__ hs) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hits) return;
    double* s = hs + h * PD_HS;
    const int64_t s0 = seg_start[h];
    const int ns = hit_nseg[h];
    const double* t = segs + 7 * total_segs;  // chief lengths
    const double* nn = segs + 6 * total_segs;  // chief refractive indices
    // point_on_beam (Beam.jl:177-205): temp = length(parent); temp += length(ray) for every ray but the last
    double temp = s[0];
    for (int k = 0; k + 1 < ns; ++k) {
        temp += t[s0 + k];
        cum[s0 + k] = temp;
    }
    cum[s0 + ns - 1] = kinf();
    // length(beam) = length_rays + length_parent (Beam.jl:125-130, :157-166); optical_path_length (Beam.jl:137-149)
    double l = 0.0, opl = s[6];
    for (int k = 0; k < ns; ++k) {
        l += t[s0 + k];
        opl += t[s0 + k] * nn[s0 + k];
    }
    const double len_total = l + s[0];
    s[10] = len_total;
    s[7] = len_total - t[s0 + ns - 1];                 // l0 = length(gauss) - length(ray)
    s[8] = 6.283185307179586 / s[4];                   // wave_number(λ) = 2π / λ
    s[9] = (opl - len_total) / s[4] * 6.283185307179586;  // ref_ϕ = Δl / λ * 2π
}

// point_on_beam(gauss.chief, z) (Beam.jl:177-205) on the compact segment table of beamlet h, and the chief / waist / divergence rays
// of the segment it selects: what gauss_parameters(gauss, z) (Gaussian.jl:298-353) starts from.
__device__ __forceinline__ RayS pd_ray_of(const double* __restrict__ segs, int64_t total_segs, int b, int64_t seg) {
    RayS r;
    const double* q = segs + (int64_t)(8 * b) * total_segs + seg;
    r.pos = {q[0], q[total_segs], q[2 * total_segs]};
    r.dir = {q[3 * total_segs], q[4 * total_segs], q[5 * total_segs]};
    r.n = q[6 * total_segs];
    return r;
}
__device__ __forceinline__ void pd_locate(double z, int64_t s0, int ns, double l_parent, const double* __restrict__ segs, const double* __restrict__ cum,
                                          int64_t total_segs, const RayS& c_last, RayS& cr, RayS& wr, RayS& dr, d3& p0) {
    const int64_t last = s0 + ns - 1;
    int64_t seg = last;
    if (ns > 1 && z < cum[last - 1]) {  // first ray (but the last) whose cumulative length exceeds z
        seg = s0;
        while (!(z < cum[seg])) ++seg;  // terminates: z < cum[last - 1]
        const double len = segs[7 * total_segs + seg];
        const double bb = cum[seg] - z;
        const RayS c = pd_ray_of(segs, total_segs, 0, seg);
        p0 = axpy3(c.pos, len - bb, c.dir);
    } else {
        const double temp = ns > 1 ? cum[last - 1] : l_parent;
        p0 = axpy3(c_last.pos, z - temp, c_last.dir);
    }
    cr = seg == last ? c_last : pd_ray_of(segs, total_segs, 0, seg);
    wr = pd_ray_of(segs, total_segs, 1, seg);
    dr = pd_ray_of(segs, total_segs, 2, seg);
}

struct PdGeom {
    double p[3];   // position(shape(pd))
    double ox[3];  // T[k,1] = orientation[1,k]: the x step in world coordinates (Photodetector.jl:76, :91-95)
    double oy[3];  // T[k,3] = orientation[3,k]
};

__global__ __launch_bounds__(256) void pd_field_kernel(int64_t n_hits, int64_t hits_per_split, const int32_t* __restrict__ seg_start,
                                                       const int32_t* __restrict__ hit_nseg, int64_t total_segs, const double* __restrict__ segs,
                                                       const double* __restrict__ cum, const double* __restrict__ hs, PdGeom G,
                                                       const double* __restrict__ xs, const double* __restrict__ ys, int32_t nx, int32_t ny,
                                                       double2* __restrict__ partial) {
    const int64_t n_pts = (int64_t)nx * ny;
    const int64_t pt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pt >= n_pts) return;
    const int i = (int)(pt % nx), j = (int)(pt / nx);
    const double x = xs[i], y = ys[j];
    const d3 p1{G.ox[0] * x + G.oy[0] * y + G.p[0], G.ox[1] * x + G.oy[1] * y + G.p[1], G.ox[2] * x + G.oy[2] * y + G.p[2]};
    const int64_t h0 = (int64_t)blockIdx.y * hits_per_split;
    const int64_t h1 = h0 + hits_per_split < n_hits ? h0 + hits_per_split : n_hits;
    double fre = 0.0, fim = 0.0;
    for (int64_t h = h0; h < h1; ++h) {  // wave-uniform: every lane walks the same beamlets (broadcast loads)
        const double* s = hs + h * PD_HS;
        const int64_t s0 = seg_start[h];
        const int ns = hit_nseg[h];
        const int64_t last = s0 + ns - 1;
        const RayS c_last = pd_ray_of(segs, total_segs, 0, last);
        // projection of the detector point on the optical axis of the last chief ray
        const d3 dp = sub3(p1, c_last.pos);
        const double l1 = dot3(dp, c_last.dir);
        const d3 p2 = axpy3(c_last.pos, l1, c_last.dir);
        const double r = norm3(sub3(p1, p2));
        const double z = s[7] + l1;
        RayS cr, wr, dr;
        d3 p0;
        pd_locate(z, s0, ns, s[0], segs, cum, total_segs, c_last, cr, wr, dr, p0);
        double w, R, psi, w0;
        gauss_parameters_at(cr, wr, dr, p0, s[4], w, R, psi, w0);
        // electric_field(gauss, r, z) Gaussian.jl:381-392
        const double ratio = s[1] / w0;                                  // beam_waist(gauss) / w0
        cx E{s[2] * ratio, s[3] * ratio};                                // E0 = electric_field(gauss) * ratio
        const double kk = s[8];
        // electric_field(r, z, E0, w0, w, k, ψ, R) OpticUtils.jl:87-89: E0 * w0 / w * exp(-r^2 / w^2) * exp(im * (k*z + ψ + (k*r^2*R)/2))
        E = cx{E.re * w0, E.im * w0};
        E = cx{E.re / w, E.im / w};
        const double ga = exp(-(r * r) / (w * w));
        E = cx{E.re * ga, E.im * ga};
        const double ph = kk * z + psi + (kk * (r * r) * R) / 2;
        double sn, cs;
        sincos(ph, &sn, &cs);
        E = cmul(E, cx{cs, sn});
        sincos(s[9], &sn, &cs);                                          // * exp(im * ref_ϕ)
        E = cmul(E, cx{cs, sn});
        const double sp = sqrt(s[5]);                                    // * sqrt(proj)
        fre += E.re * sp;
        fim += E.im * sp;
    }
    partial[(int64_t)blockIdx.y * n_pts + pt] = make_double2(fre, fim);
}

__global__ void pd_reduce_kernel(const double2* __restrict__ partial, int32_t n_splits, int64_t n_pts, double2* __restrict__ field) {
    const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= n_pts) return;
    double re = 0.0, im = 0.0;
    for (int s = 0; s < n_splits; ++s) {
        const double2 v = partial[(int64_t)s * n_pts + pt];
        re += v.x;
        im += v.y;
    }
    field[pt] = make_double2(field[pt].x + re, field[pt].y + im);
}

// gauss_parameters(gauss, z) for one beamlet (slot 0 of the per-hit tables) at n values of z: the same gather / prepare / locate /
// gauss_parameters_at sequence the Photodetector field runs per grid point
__global__ void gp_pick_kernel(int64_t canon, const int32_t* __restrict__ order, const int32_t* __restrict__ nseg, const double* __restrict__ aux,
                               const double* __restrict__ lambda, int32_t* __restrict__ hit_nseg, int32_t* __restrict__ node_hit, double* __restrict__ hs,
                               PdPrefix pre) {
    const int32_t nd = order[canon];
    hit_nseg[0] = nseg[nd] + pre.len(nd);
    node_hit[nd] = 0;
    hs[0] = aux[(int64_t)nd * 4 + 0];
    hs[1] = aux[(int64_t)nd * 4 + 1];
    hs[2] = aux[(int64_t)nd * 4 + 2];
    hs[3] = aux[(int64_t)nd * 4 + 3];
    hs[4] = lambda[nd];
    hs[5] = 1.0;
}
__global__ void gp_eval_kernel(int32_t n, const double* __restrict__ zs, int ns, int64_t total_segs, const double* __restrict__ segs, const double* __restrict__ cum,
                               const double* __restrict__ hs, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RayS c_last = pd_ray_of(segs, total_segs, 0, ns - 1);
    RayS cr, wr, dr;
    d3 p0;
    pd_locate(zs[i], 0, ns, hs[0], segs, cum, total_segs, c_last, cr, wr, dr, p0);
    double w, R, psi, w0;
    gauss_parameters_at(cr, wr, dr, p0, hs[4], w, R, psi, w0);
    out[4 * i + 0] = w;
    out[4 * i + 1] = R;
    out[4 * i + 2] = psi;
    out[4 * i + 3] = w0;
}

}  // namespace

static PdPrefix prefix_of(const bmo_trace_result* res) {
    PdPrefix p{nullptr, nullptr, nullptr, 0, 0};
    if (res->pre_start.p) p = PdPrefix{(const int32_t*)res->pre_start.p, (const double*)res->pre_segs.p, (const double*)res->pre_opl.p, res->pre_roots, res->pre_total};
    return p;
}

extern "C" int bmo_result_set_gauss_prefix(bmo_trace_result* res, int64_t n_roots, const int32_t* prefix_start, const double* prefix_segs, const double* opl_parent) {
    if (!res || !prefix_start || !opl_parent || n_roots <= 0) return fail(BMO_ERR_INVALID, "bmo_result_set_gauss_prefix: bad argument");
    if (res->kind != BMO_BEAM_GAUSSIAN) return fail(BMO_ERR_INVALID, "bmo_result_set_gauss_prefix: not a GaussianBeamlet solution");
    if (n_roots != res->n_roots) return fail(BMO_ERR_INVALID, "bmo_result_set_gauss_prefix: one entry per root beamlet of the solution is expected");
    if (prefix_start[0] != 0) return fail(BMO_ERR_INVALID, "bmo_result_set_gauss_prefix: prefix_start[0] must be 0");
    for (int64_t i = 0; i < n_roots; ++i)
        if (prefix_start[i + 1] < prefix_start[i]) return fail(BMO_ERR_INVALID, "bmo_result_set_gauss_prefix: prefix_start must not decrease");
    const int64_t total = prefix_start[n_roots];
    if (total > 0 && !prefix_segs) return fail(BMO_ERR_INVALID, "bmo_result_set_gauss_prefix: prefix_segs missing");
    HIP_TRY(hipSetDevice(res->device));
    int rc;
    res->pre_start.release();
    res->pre_segs.release();
    res->pre_opl.release();
    if ((rc = res->pre_start.alloc((size_t)(n_roots + 1) * 4)) || (rc = res->pre_segs.alloc((size_t)std::max<int64_t>(total, 1) * PD_SEG_PLANES * 8)) ||
        (rc = res->pre_opl.alloc((size_t)n_roots * 8)))
        return rc;
    HIP_TRY(hipMemcpy(res->pre_start.p, prefix_start, (size_t)(n_roots + 1) * 4, hipMemcpyHostToDevice));
    if (total > 0) HIP_TRY(hipMemcpy(res->pre_segs.p, prefix_segs, (size_t)total * PD_SEG_PLANES * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(res->pre_opl.p, opl_parent, (size_t)n_roots * 8, hipMemcpyHostToDevice));
    res->pre_roots = n_roots;
    res->pre_total = total;
    return BMO_OK;
}

extern "C" int bmo_gauss_parameters(bmo_trace_result* res, int64_t node, const double* zs, int32_t n, double* out) {
    if (!res || !zs || !out || n <= 0) return fail(BMO_ERR_INVALID, "bmo_gauss_parameters: bad argument");
    if (res->kind != BMO_BEAM_GAUSSIAN) return fail(BMO_ERR_INVALID, "bmo_gauss_parameters: not a GaussianBeamlet solution");
    if (node < 0 || node >= res->n_nodes) return fail(BMO_ERR_INVALID, "bmo_gauss_parameters: bad beamlet index");
    if (!res->has_log) return fail(BMO_ERR_INVALID, "bmo_gauss_parameters: the solution was solved with record_segments = 0 (no segments to evaluate)");
    HIP_TRY(hipSetDevice(res->device));
    int rc;
    DevBuf hit_nseg, node_hit, seg_start, hs, segs, cum, d_z, d_out;
    if ((rc = hit_nseg.alloc(4)) || (rc = node_hit.alloc((size_t)res->n_nodes * 4)) || (rc = seg_start.alloc(4)) || (rc = hs.alloc(PD_HS * 8)) ||
        (rc = d_z.alloc((size_t)n * 8)) || (rc = d_out.alloc((size_t)n * 32)))
        return rc;
    hipStream_t st = 0;
    HIP_TRY(hipMemsetAsync(node_hit.p, 0xFF, (size_t)res->n_nodes * 4, st));
    HIP_TRY(hipMemsetAsync(hs.p, 0, PD_HS * 8, st));
    HIP_TRY(hipMemsetAsync(seg_start.p, 0, 4, st));
    hipLaunchKernelGGL(gp_pick_kernel, dim3(1), dim3(1), 0, st, node, (const int32_t*)res->order.p, (const int32_t*)res->n_nseg.p, (const double*)res->n_aux.p,
                       (const double*)res->n_lambda.p, (int32_t*)hit_nseg.p, (int32_t*)node_hit.p, (double*)hs.p, prefix_of(res));
    int32_t ns = 0;
    HIP_TRY(hipMemcpyAsync(&ns, hit_nseg.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ns <= 0) return fail(BMO_ERR_INTERNAL, "bmo_gauss_parameters: beamlet without segments");
    if ((rc = segs.alloc((size_t)ns * PD_SEG_PLANES * 8)) || (rc = cum.alloc((size_t)ns * 8))) return rc;
    for (const Chunk& c : res->chunks)
        if (c.count > 0)
            hipLaunchKernelGGL(pd_gather_kernel, dim3((unsigned)((c.count + 255) / 256)), dim3(256), 0, st, c, (const int32_t*)node_hit.p,
                               (const int32_t*)seg_start.p, (int64_t)ns, (double*)segs.p, (double*)hs.p, prefix_of(res));
    hipLaunchKernelGGL(pd_prepare_kernel, dim3(1), dim3(256), 0, st, (int64_t)1, (const int32_t*)seg_start.p, (const int32_t*)hit_nseg.p, (int64_t)ns,
                       (const double*)segs.p, (double*)cum.p, (double*)hs.p);
    HIP_TRY(hipMemcpyAsync(d_z.p, zs, (size_t)n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gp_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const double*)d_z.p, (int)ns, (int64_t)ns, (const double*)segs.p,
                       (const double*)cum.p, (const double*)hs.p, (double*)d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n * 32, hipMemcpyDeviceToHost));
    return BMO_OK;
}

extern "C" int bmo_photodetector_field(bmo_trace_result* res, int32_t detector, const double position[3], const double orientation[9], const double* xs,
                                       const double* ys, int32_t nx, int32_t ny, double* field_inout, double* kernel_ms) {
    if (!res || !position || !orientation || !xs || !ys || !field_inout || nx <= 0 || ny <= 0) return fail(BMO_ERR_INVALID, "bmo_photodetector_field: bad argument");
    if (detector < 0 || detector >= res->n_detectors) return fail(BMO_ERR_INVALID, "bmo_photodetector_field: bad detector slot");
    if (kernel_ms) *kernel_ms = 0.0;
    if (res->kind != BMO_BEAM_GAUSSIAN) return BMO_OK;  // other beams leave no record (Photodetector.jl:57-60)
    const int64_t rows = res->det_count[detector];
    if (rows % 3) return fail(BMO_ERR_INVALID, "bmo_photodetector_field: slot does not hold photodetector records");
    const int64_t H = rows / 3;
    if (H == 0) return BMO_OK;
    if (!res->has_log) return fail(BMO_ERR_INVALID, "bmo_photodetector_field: the solution was solved with record_segments = 0 (gauss_parameters needs the beamlets' segments)");
    HIP_TRY(hipSetDevice(res->device));
    const int64_t nn = res->n_nodes, n_pts = (int64_t)nx * ny;
    int rc;
    DevBuf hit_node, hit_nseg, node_hit, seg_start, hs, tmp, segs, cum, d_xs, d_ys, partial, d_field;
    if ((rc = hit_node.alloc((size_t)H * 4)) || (rc = hit_nseg.alloc((size_t)H * 4)) || (rc = node_hit.alloc((size_t)nn * 4)) ||
        (rc = seg_start.alloc((size_t)H * 4)) || (rc = hs.alloc((size_t)H * PD_HS * 8)))
        return rc;
    hipStream_t st = 0;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    HIP_TRY(hipMemsetAsync(node_hit.p, 0xFF, (size_t)nn * 4, st));
    HIP_TRY(hipMemsetAsync(hs.p, 0, (size_t)H * PD_HS * 8, st));
    const unsigned hb = (unsigned)((H + 255) / 256);
    hipLaunchKernelGGL(pd_hit_nodes_kernel, dim3(hb), dim3(256), 0, st, (const int32_t*)res->det_node.p, res->det_offset[detector], H,
                       (const int32_t*)res->order.p, (const int32_t*)res->n_nseg.p, (const double*)res->n_aux.p, (const double*)res->n_lambda.p,
                       (const double*)res->det_data.p, (int32_t*)hit_node.p, (int32_t*)hit_nseg.p, (int32_t*)node_hit.p, (double*)hs.p, prefix_of(res));
    size_t tmp_bytes = 0;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (const int32_t*)hit_nseg.p, (int32_t*)seg_start.p, (int)H, st));
    if ((rc = tmp.alloc(tmp_bytes))) return rc;
    HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp_bytes, (const int32_t*)hit_nseg.p, (int32_t*)seg_start.p, (int)H, st));
    int32_t last_start = 0, last_n = 0;
    HIP_TRY(hipMemcpyAsync(&last_start, (const int32_t*)seg_start.p + H - 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&last_n, (const int32_t*)hit_nseg.p + H - 1, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t total_segs = (int64_t)last_start + last_n;
    // workgroups: 256 points each x beamlet ranges, enough ranges to fill the chip when the grid is small
    const unsigned pt_blocks = (unsigned)((n_pts + 255) / 256);
    int64_t n_splits = (2048 + pt_blocks - 1) / pt_blocks;
    if (n_splits > H) n_splits = H;
    if (n_splits > 65535) n_splits = 65535;
    if (n_splits < 1) n_splits = 1;
    const int64_t hits_per_split = (H + n_splits - 1) / n_splits;
    n_splits = (H + hits_per_split - 1) / hits_per_split;
    if ((rc = segs.alloc((size_t)total_segs * PD_SEG_PLANES * 8)) || (rc = cum.alloc((size_t)total_segs * 8)) || (rc = d_xs.alloc((size_t)nx * 8)) ||
        (rc = d_ys.alloc((size_t)ny * 8)) || (rc = partial.alloc((size_t)n_splits * n_pts * 16)) || (rc = d_field.alloc((size_t)n_pts * 16)))
        return rc;
    for (const Chunk& c : res->chunks)
        if (c.count > 0)
            hipLaunchKernelGGL(pd_gather_kernel, dim3((unsigned)((c.count + 255) / 256)), dim3(256), 0, st, c, (const int32_t*)node_hit.p,
                               (const int32_t*)seg_start.p, total_segs, (double*)segs.p, (double*)hs.p, prefix_of(res));
    hipLaunchKernelGGL(pd_prepare_kernel, dim3(hb), dim3(256), 0, st, H, (const int32_t*)seg_start.p, (const int32_t*)hit_nseg.p, total_segs,
                       (const double*)segs.p, (double*)cum.p, (double*)hs.p);
    HIP_TRY(hipMemcpyAsync(d_xs.p, xs, (size_t)nx * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_ys.p, ys, (size_t)ny * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_field.p, field_inout, (size_t)n_pts * 16, hipMemcpyHostToDevice, st));
    PdGeom G;
    for (int k = 0; k < 3; ++k) {
        G.p[k] = position[k];
        G.ox[k] = orientation[0 * 3 + k];  // T[k,1] with T = transpose(orientation)
        G.oy[k] = orientation[2 * 3 + k];  // T[k,3]
    }
    hipLaunchKernelGGL(pd_field_kernel, dim3(pt_blocks, (unsigned)n_splits), dim3(256), 0, st, H, hits_per_split, (const int32_t*)seg_start.p,
                       (const int32_t*)hit_nseg.p, total_segs, (const double*)segs.p, (const double*)cum.p, (const double*)hs.p, G,
                       (const double*)d_xs.p, (const double*)d_ys.p, nx, ny, (double2*)partial.p);
    hipLaunchKernelGGL(pd_reduce_kernel, dim3(pt_blocks), dim3(256), 0, st, (const double2*)partial.p, (int32_t)n_splits, n_pts, (double2*)d_field.p);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (kernel_ms) *kernel_ms = ms;
    HIP_TRY(hipMemcpy(field_inout, d_field.p, (size_t)n_pts * 16, hipMemcpyDeviceToHost));
    return BMO_OK;
}
