"""Host half of the model plug-in surface (Model::UID / N_EQ / Parameters / SP_TYPE / rhs).

Mirrors reference src/models/model_204.hpp:15-115.  ``rhs`` here is the HOST evaluation
the reference driver uses for its initial-step guess (main.cpp:615-631); the device
evaluation is csrc/hlm_models.h.
"""
from dataclasses import dataclass, astuple

import numpy as np

from ._lib import UID_TOY3, UID_204, UID_DUMMY

# I_O/parameters_loader.hpp:19-37 -- 2 x long + 15 x double = 136 B, field order is ABI
SP_DTYPE = np.dtype(
    [("stream", "<i8"), ("next_stream", "<i8")]
    + [(n, "<f8") for n in ("c1", "infil", "perco", "Hu", "lat", "sw", "ss", "n_mann", "slope",
                            "L", "A_h", "alpha3", "alpha4", "melt_f", "temp_thr")]
)
assert SP_DTYPE.itemsize == 136


@dataclass
class Parameters:
    """Model::Parameters, model_204.hpp:22-30 (same defaults)."""
    initialStep: float = 0.01
    rtol: float = 1e-6
    atol: float = 1e-9
    safety: float = 0.9
    minScale: float = 0.2
    maxScale: float = 10.0

    def as_tuple(self):
        return tuple(float(x) for x in astuple(self))


class Model204:
    """5-equation snow / static / surface / gravitational / aquifer runoff model."""
    UID = UID_204
    N_EQ = 5
    SP_TYPE = SP_DTYPE
    Parameters = Parameters
    STATE_NAMES = ("h_snow", "h_static", "h_surface", "h_grav", "h_aquifer")

    @staticmethod
    def rhs(t, y, sp, F=()):
        """model_204.hpp:43-114 on the host (float64).  ``sp`` is one SP_DTYPE record."""
        h_snow, h_stat, h_surf, h_grav, h_aq = (float(v) for v in y)
        rain = float(np.float32(F[0])) if len(F) > 0 else 0.0
        temp = float(np.float32(F[1])) if len(F) > 1 else 0.0
        melt = min(h_snow, temp * sp["melt_f"]) if temp >= sp["temp_thr"] else 0.0
        x1 = rain + melt
        d0 = rain - melt
        x2 = max(0.0, x1 + h_stat - sp["Hu"])
        d1 = x1 - x2 - (h_stat / sp["Hu"]) * min(0.1 * temp, h_stat)
        x3 = min(x2, sp["infil"])
        with np.errstate(invalid="ignore"):
            p = float(np.power(np.float64(h_surf), 2.0 / 3.0))
        alfa2 = (1.0 / sp["n_mann"]) * p * np.sqrt(sp["slope"])
        wv = alfa2 * sp["L"] / sp["A_h"] * 60.0
        w = 1.0 if np.isnan(wv) else min(1.0, wv)  # fmin(1.0, NaN) == 1.0
        d2 = (x2 - x3) - h_surf * w
        x4 = min(x3, sp["perco"])
        d3 = (x3 - x4) - (h_grav / sp["alpha3"] if sp["alpha3"] >= 1.0 else 0.0)
        d4 = x4 - (h_aq / sp["alpha4"] if sp["alpha4"] >= 1.0 else 0.0)
        return np.array([d0, d1, d2, d3, d4], dtype=np.float64)


class DummyModel:
    """The README's 5-equation linear test model (README.md:41-42,90); RHS from
    model_dummy_python.ipynb:65-89.  Its source is absent from the reference tree; the
    build assigns UID 0."""
    UID = UID_DUMMY
    N_EQ = 5
    SP_TYPE = SP_DTYPE
    Parameters = Parameters
    STATE_NAMES = ("Var0", "Var1", "Var2", "Var3", "Var4")

    @staticmethod
    def rhs(t, y, sp=None, F=()):
        H0, H1, _, H3, _ = (float(v) for v in y)
        Y0, X2, I2, I3 = 0.5 * H0, 0.3 * H1, 0.6 * H1, 0.4 * H3
        return np.array([1.0 - Y0, 1.2 + Y0 - X2 - 0.4 - I2, X2 - 0.2, I2 - I3 - 0.3, I3 - 0.1])


class ToyModel3:
    """The build's "Add a New Model" example (reference README.md, plug-in section): three stores, one
    forcing, a feedback from the last store to the first (csrc/hlm_models.h: ToyDev;
    include/hlm/models/model_toy3.hpp).  k1 = 1/alpha3 and k3 = 1/alpha4 come from the SpatialParams
    record when one is given, else 0.05 and 0.01."""
    UID = UID_TOY3
    N_EQ = 3
    SP_TYPE = SP_DTYPE
    Parameters = Parameters
    STATE_NAMES = ("store0", "store1", "store2")

    @staticmethod
    def rhs(t, y, sp=None, F=()):
        y0, y1, y2 = (float(v) for v in y)
        k1 = 1.0 / float(sp["alpha3"]) if sp is not None and sp["alpha3"] > 0 else 0.05
        k3 = 1.0 / float(sp["alpha4"]) if sp is not None and sp["alpha4"] > 0 else 0.01
        inflow = float(np.float32(F[0])) if len(F) > 0 else 0.0
        out0, up, out1 = k1 * y0, 0.3 * y0 * y2, 0.02 * y1 * y1
        return np.array([inflow - out0 - up, out0 - out1, out1 - k3 * y2 + up])


MODELS = {Model204.UID: Model204, DummyModel.UID: DummyModel, ToyModel3.UID: ToyModel3}


def initial_step_guess(model, y0_row, sp_row, F=(), rtol=1e-6, atol=1e-9):
    """SciPy-style h0 of the reference driver, main.cpp:615-631.  NB the reference
    evaluates it at y = 0 (so d0 = 0 and the guess is always 1e-6); pass y0_row=zeros to
    reproduce that, or the real initial state for a meaningful guess."""
    y = np.asarray(y0_row, dtype=np.float64)
    f0 = model.rhs(0.0, y, sp_row, F)
    scale = atol + rtol * np.abs(y)
    d0 = np.sqrt(np.sum((y / scale) ** 2))
    d1 = np.sqrt(np.sum((f0 / scale) ** 2))
    return max(1e-6, 0.01 * d0 / (d1 + 1e-16))
