"""Parameter-CSV and lookup-CSV readers (drop-in formats).

``load_spatial_params`` mirrors reference src/I_O/parameters_loader.cpp:8-107: columns
are located by header name, the same required set, the same unit conversions, the same
error conditions (raised as RuntimeError where the reference throws std::runtime_error).
``LookupMapper`` mirrors src/I_O/forcing_loader.cpp:14-64.
"""
import numpy as np

from .models import SP_DTYPE

REQUIRED_COLUMNS = (  # parameters_loader.cpp:35-49
    "stream", "next_stream", "i2", "i3", "hu", "centroid_lat", "sw", "ss", "n", "slope",
    "length_km", "drainage_area_km2", "melt", "t_thres", "res_ss", "res_gw")

C_1 = 0.001 / 60.0  # mm/hr -> m/min, parameters_loader.cpp:57


def load_spatial_params(csv_path):
    try:
        f = open(csv_path, "r")
    except OSError:
        raise RuntimeError("Failed to open parameter file: " + str(csv_path))
    with f:
        header = f.readline()
        if header == "":
            raise RuntimeError("Empty parameter file: " + str(csv_path))
        headers = header.rstrip("\n").rstrip("\r").split(",")
        idx = {name: i for i, name in enumerate(headers)}
        for name in REQUIRED_COLUMNS:
            if name not in idx:
                raise RuntimeError("Missing column '" + name + "' in " + str(csv_path))
        rows = []
        for line in f:
            line = line.rstrip("\n").rstrip("\r")
            if line == "":
                continue  # parameters_loader.cpp:62
            fields = line.split(",")
            if len(fields) < len(headers):
                raise RuntimeError("Bad row with too few fields in " + str(csv_path))
            rows.append(fields)
    out = np.zeros(len(rows), dtype=SP_DTYPE)
    g = lambda fields, name: float(fields[idx[name]])
    for k, fields in enumerate(rows):
        r = out[k]
        r["stream"] = int(fields[idx["stream"]])
        r["next_stream"] = int(fields[idx["next_stream"]])
        r["c1"] = C_1
        r["Hu"] = g(fields, "hu")
        r["lat"] = g(fields, "centroid_lat")
        r["sw"] = g(fields, "sw")
        r["ss"] = g(fields, "ss")
        r["n_mann"] = g(fields, "n")
        r["slope"] = g(fields, "slope")
        r["L"] = g(fields, "length_km")
        r["A_h"] = g(fields, "drainage_area_km2")
        r["melt_f"] = g(fields, "melt")
        r["temp_thr"] = g(fields, "t_thres")
        r["infil"] = g(fields, "i2") * C_1            # :92-95
        r["perco"] = g(fields, "i3") * C_1
        r["alpha3"] = g(fields, "res_ss") * 24.0 * 60.0  # days -> minutes, :98-101
        r["alpha4"] = g(fields, "res_gw") * 24.0 * 60.0
    return out


class LookupMapper:
    """stream id -> (lat_index, lon_index); header skipped, missing id -> (-1, -1)
    (forcing_loader.cpp:54-60)."""

    def __init__(self, path):
        self.path = path
        self.map = {}

    def load(self):
        try:
            with open(self.path, "r") as f:
                f.readline()
                for line in f:
                    parts = line.strip().split(",")
                    if len(parts) < 3:
                        continue
                    self.map[int(parts[0])] = (int(parts[1]), int(parts[2]))
        except OSError:
            return False
        return True

    def get_lat_lon(self, stream_id):
        return self.map.get(int(stream_id), (-1, -1))
