"""Host side of corner grinding: the driver's selection rule and the `grind` decision, with the uniform
draws handed in by the caller so that a given random stream reproduces the reference exactly.

  keep_mask      Subzero.m:339-347   which floes are left alone this round
  grind_flags    corners.m:70-71,91-94   per selected floe: which vertices break off, and whether the floe is reshaped

The per-vertex inputs (interior angles, the `da` flags) come from the GPU: capi.FloeInteractions.corner_candidates
(szCornerCandidates).  frac_corner.m -- cutting the corners off with polyshape and making the new floes -- reshapes the
floe array and stays on the MATLAB side; afterwards the array goes back with szFloesResync.
"""
import numpy as np


def keep_mask(u, overlap_area, area, Nb, min_floe_size=None):
    """keep = rand(N,1) > 0.7; keep(OverlapArea./area > 0.15) = 0; keep(1:Nb) = 1; [KEEP_MIN: keep(area < min) = 1]
    (Subzero.m:341-347).  u: the N uniform draws.  Returns the boolean keep; Floe(~keep) goes to corners."""
    u = np.asarray(u, np.float64)
    keep = u > 0.7
    keep[np.asarray(overlap_area) / np.asarray(area) > 0.15] = False
    keep[:Nb] = True
    if min_floe_size is not None:
        keep[np.asarray(area) < min_floe_size] = True
    return keep


def draws_needed(count):
    """How many uniform draws corners.m:71 consumes, floe by floe: one per vertex of every eligible floe."""
    return int(np.asarray(count).sum())


def grind_flags(offsets, count, angles, da, u):
    """corners.m:70-71,91-94.  u: draws_needed(count) uniform draws, consumed in the order of the selected floes.
    Returns (grind, fracture): grind[offsets[q] : offsets[q] + count[q]] the vertices of floe q to break off,
    fracture[q] whether sum(grind) > 1 (the floe goes to frac_corner)."""
    u = np.asarray(u, np.float64)
    grind = np.zeros(len(angles), bool)
    fracture = np.zeros(len(count), bool)
    at = 0
    for q, n in enumerate(count):
        if n == 0:
            continue
        o = offsets[q]
        anorm = 180.0 - 360.0 / n
        break1 = u[at:at + n] > angles[o:o + n] / anorm
        at += n
        g = break1 & (da[o:o + n] != 0)
        grind[o:o + n] = g
        fracture[q] = g.sum() > 1
    return grind, fracture
