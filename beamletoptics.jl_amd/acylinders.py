"""Acylindric lens surfaces (host-side builders), mirror of src/SDFs/AcylindricalSDF.jl.

Kept in its own module; `shapes` pulls these names in so `bmo_amd.AcylindricalSurface` etc. work like the reference exports.
"""
import math

from . import linalg as la

K_ACYL_CONVEX, K_ACYL_CONCAVE = 17, 18


def make(sh):
    """Create the classes against the `shapes` module (avoids a circular import)."""

    class _AcylinderSDF(sh.AbstractSDF):
        flags = sh.FLAG_INEXACT  # first-order distance estimate (same 2D profile as the aspheres)

        def __init__(self, radius, diameter, height, conic_constant, coefficients):
            super().__init__()
            self.radius, self.diameter, self.height = float(radius), float(diameter), float(height)
            self.conic_constant = float(conic_constant)
            self.coefficients = [float(x) for x in coefficients]
            self.max_sag = sh.max_aspheric_value(1 / self.radius, self.conic_constant, self.coefficients, self.diameter)

        def edge_sag_value(self):
            return sh.aspheric_equation(self.diameter / 2, 1 / self.radius, self.conic_constant, self.coefficients)

        def params(self):
            return [self.radius, self.diameter, self.height, self.conic_constant, self.max_sag[0]]

        def slope_bound(self):
            """See _AsphericalSurfaceSDF.slope_bound: the same 2D profile, extruded instead of revolved; the extrusion combines the
            profile term d2 with the exact slab term |x| - h/2 by max / norm(max.(., 0)) (AcylindricalSDF.jl:55-74), which keeps
            sdf >= dist / K for K >= 1."""
            return sh._profile_slope_bound(self.radius, self.conic_constant, self.coefficients, self.diameter)

        def _local_bound(self):
            zs = [0.0, self.edge_sag_value(), self.max_sag[0]]
            lo, hi = min(zs), max(zs)
            return [0, (lo + hi) / 2, 0], math.sqrt(((hi - lo) / 2) ** 2 + (self.diameter / 2) ** 2 + (self.height / 2) ** 2)

    class AconvexCylinderSDF(_AcylinderSDF):  # AcylindricalSDF.jl:16-74
        kind = K_ACYL_CONVEX

        @property
        def thickness(self):
            return abs(self.edge_sag_value())

    class AconcaveCylinderSDF(_AcylinderSDF):  # AcylindricalSDF.jl:83-141
        kind = K_ACYL_CONCAVE

        @property
        def thickness(self):
            sag = self.edge_sag_value()
            return abs(sag) if (self.max_sag[0] > 0 and sag < 0) else 0.0

    class AcylindricalSurface(sh.CylindricalSurface):  # AcylindricalSDF.jl:164-239
        def __init__(self, radius, diameter, height, conic_constant, coefficients, mechanical_diameter=None):
            super().__init__(radius, diameter, height, mechanical_diameter)
            self.conic_constant = float(conic_constant)
            self.coefficients = [float(x) for x in coefficients]

    def surface_sdf(s, orient):  # AcylindricalSDF.jl:207-227
        args = (s.diameter, s.height, s.conic_constant, s.coefficients)
        if orient == "forward":
            return AconvexCylinderSDF(s.radius, *args) if s.radius > 0 else AconcaveCylinderSDF(s.radius, *args)
        if orient == "backward":
            return AconcaveCylinderSDF(s.radius, *args) if s.radius > 0 else AconvexCylinderSDF(-s.radius, *args)
        raise ValueError(orient)

    def edge_sag(s):  # AcylindricalSDF.jl:197-205
        return sh.aspheric_equation(s.diameter / 2, 1 / s.radius, s.conic_constant, s.coefficients)

    return AconvexCylinderSDF, AconcaveCylinderSDF, AcylindricalSurface, surface_sdf, edge_sag
