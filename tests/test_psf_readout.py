"""PSF read-out: intensity(psf) PSFDetector.jl:190-237.

CPU part: the oracle restatement reproduces the reference's Airy-disc KAT (test/runtests.jl:2764-2802).
GPU part: bmo_psf_intensity against the oracle on the same hits (tolerance: the GPU sums hits in 256-hit blocks, so the
comparison is relative to the peak field magnitude; 1e-10 as BASELINE.json's north_star states for FP64).
"""
import math

import numpy as np
import pytest

import bmo_amd as bmo

RTOL = 1e-10


def airy_setup(num_rays=1000):
    # runtests.jl:2764-2784: almost thin plano-convex lens, plane wave, detector at the paraxial focus + 0.13 mm
    l, R1, R2, d, n, lam, D = 1e-3, 100e-3, math.inf, 25.4e-3, 1.5, 1e-6, 15e-3
    cs = bmo.UniformDiscSource([0, -10e-3, 0], [0, 1, 0], D, lam, num_rays=num_rays, e1=[1, 0, 0])
    lens = bmo.SphericalLens(R1, R2, l, d, lambda lam_: n)
    psfd = bmo.PSFDetector(10e-3)
    bmo.translate3d(psfd, [0, 200e-3 + 0.13e-3, 0])
    return bmo.System([lens, psfd]), cs, psfd, lam, D


def test_linrange_matches_julia_lerpi():
    xs = bmo.linalg.linrange(-1.0, 2.0, 4)
    assert xs.tolist() == [-1.0, (1 - 1 / 3) * -1.0 + (1 / 3) * 2.0, (1 - 2 / 3) * -1.0 + (2 / 3) * 2.0, 2.0]
    assert bmo.linalg.linrange(3.0, 5.0, 1).tolist() == [3.0]


def test_airy_disc_kat_oracle(oracle):
    """runtests.jl:2786-2802: first zero of the numerical PSF through the peak == 1.22 lambda f / D within 1 %."""
    system, cs, psfd, lam, D = airy_setup()
    oracle.solve_system(system, cs)
    assert len(psfd.data) == 1000  # runtests.jl:2806
    xs, zs, I = psfd.intensity(n=500, crop_factor=5, center="bbox",
                               _intensity_fn=lambda *a: oracle.psf_intensity(*a)[0])
    ix, jx = np.unravel_index(np.argmax(I), I.shape)
    num_min = xs[np.argmin(I[:, jx])]
    airy_min = 1.22 * lam * 200e-3 / D
    assert abs(abs(num_min) - airy_min) <= 1e-2 * airy_min
    psfd.empty()
    assert len(psfd.data) == 0


def test_psf_data_fields(oracle):
    """PSFData (PSFDetector.jl:77-89): hit on the detector plane, k = 2 pi / lambda, proj = |dir . n|, opl = sum n*l."""
    system, cs, psfd, lam, D = airy_setup(num_rays=64)
    oracle.solve_system(system, cs)
    d = psfd.data
    assert np.allclose(d[:, 1], 200e-3 + 0.13e-3, atol=1e-12)
    assert np.allclose(d[:, 8], 2 * math.pi / lam, rtol=0, atol=0)
    assert np.allclose(d[:, 7], np.abs(d[:, 4]), rtol=1e-15)  # detector normal is -y
    assert np.all(d[:, 6] > 0.21)  # > geometric path: 1 mm of glass at n = 1.5


@pytest.mark.gpu
@pytest.mark.parametrize("num_rays,n", [(1000, 500), (300, 33), (1, 7)])
def test_gpu_psf_intensity_matches_oracle(oracle, num_rays, n):
    system, cs, psfd, lam, D = airy_setup(num_rays=num_rays)
    bmo.solve_system(system, cs)  # the engine fills psfd.data
    assert len(psfd.data) == num_rays
    xs, zs = psfd.sample_axes(n=n, crop_factor=5, center="bbox") if num_rays > 1 else (bmo.linalg.linrange(-1e-5, 1e-5, n),) * 2
    o = psfd.orientation()
    I_gpu, F_gpu, ms = bmo.abi.psf_intensity(psfd.data, psfd.position(), o[:, 0], o[:, 2], xs, zs, want_field=True)
    I_ref, F_ref = oracle.psf_intensity(psfd.data, psfd.position(), o[:, 0], o[:, 2], xs, zs)
    peak = np.abs(F_ref).max()
    assert np.abs(F_gpu - F_ref).max() <= RTOL * peak
    assert np.abs(I_gpu - I_ref).max() <= 2 * RTOL * peak * peak
    assert np.array_equal(I_gpu, F_gpu.real ** 2 + F_gpu.imag ** 2)


@pytest.mark.gpu
def test_gpu_psf_intensity_device_resident_hits_and_airy_kat(oracle):
    """Same KAT as the CPU test, read out by the engine from the hit buffer that is still resident in HBM."""
    system, cs, psfd, lam, D = airy_setup()
    bundle = bmo.RayBundle.from_beams(cs.beams)
    scene = bmo.CompiledScene(system, bundle.lambdas)
    eng = bmo.Engine(scene, 0)
    dev = eng.upload(bundle)
    res = eng.trace_device(dev, 100)
    ptr, cnt = eng.result_device_hits(res, 0)
    assert cnt == 1000
    view = eng.result_view(res)
    psfd.data = view.detector_hits(0).copy()
    xs, zs = psfd.sample_axes(n=500, crop_factor=5, center="bbox")
    o = psfd.orientation()
    I, _, ms = bmo.abi.psf_intensity(None, psfd.position(), o[:, 0], o[:, 2], xs, zs, hits_device_ptr=ptr, n_hits=cnt)
    I_host, _, _ = bmo.abi.psf_intensity(psfd.data, psfd.position(), o[:, 0], o[:, 2], xs, zs)
    assert np.array_equal(I, I_host)
    ix, jx = np.unravel_index(np.argmax(I), I.shape)
    num_min = xs[np.argmin(I[:, jx])]
    airy_min = 1.22 * lam * 200e-3 / D
    assert abs(abs(num_min) - airy_min) <= 1e-2 * airy_min
    eng.free_result(res)
    eng.free_batch(dev)
    eng.close()


@pytest.mark.gpu
def test_gpu_psf_intensity_rejects_bad_arguments():
    with pytest.raises(RuntimeError):
        bmo.abi.psf_intensity(np.zeros((3, 9)), [0, 0, 0], [1, 0, 0], [0, 0, 1], [0.0, 1.0], [0.0, 1.0], device=99)
