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
