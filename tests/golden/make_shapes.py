"""Decode the reference's own fixture test/test_conservation/FloeShapes.mat (a DATA file: 462 MATLAB
polyshape objects) into tests/golden/floe_shapes.npz.

Runs only in the build container (needs /root/reference).  The .mat stores the objects MCOS-encoded
in its __function_workspace__; each polyshape carries Vertices plus MATLAB's own cached
PolygonInfo = [?, area, perimeter, centroid_x, centroid_y, xmin, ymin, xmax, ymax], which pins the
oracle's area/centroid restatement against values MATLAB itself computed.
"""
import io
import sys
import numpy as np
import scipy.io as sio
from scipy.io.matlab._mio5 import MatFile5Reader

SRC = "/root/reference/test/test_conservation/FloeShapes.mat"


def decode(path=SRC):
    m = sio.loadmat(path)
    b = m["__function_workspace__"].tobytes()
    hdr = b"MATLAB 5.0 MAT-file".ljust(116, b" ") + b"\x00" * 8 + b"\x00\x01IM"
    f = io.BytesIO(hdr + b[8:])
    r = MatFile5Reader(f, struct_as_record=False)
    r.initialize_read()
    f.seek(128)
    h, _ = r.read_var_header()
    res = r.read_var_array(h, process=False)
    cells = res[0, 0].MCOS["arr"][0]
    shapes = []
    for i in range(cells.shape[0]):
        e = cells[i, 0]
        if isinstance(e, np.ndarray) and e.dtype == object and e.size == 1 and hasattr(e[0, 0], "Vertices"):
            shapes.append(e[0, 0])
    return shapes


def main(out):
    shapes = decode()
    off = np.zeros(len(shapes) + 1, np.int64)
    xs, ys, info = [], [], []
    for k, s in enumerate(shapes):
        v = np.asarray(s.Vertices, np.float64)
        assert not np.isnan(v).any()
        xs.append(v[:, 0]); ys.append(v[:, 1])
        off[k + 1] = off[k] + v.shape[0]
        info.append(np.asarray(s.PolygonInfo, np.float64).ravel())
    np.savez_compressed(out, x=np.concatenate(xs), y=np.concatenate(ys), off=off,
                        polygon_info=np.stack(info))
    print("shapes:", len(shapes), "verts:", off[-1])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/floe_shapes.npz")
