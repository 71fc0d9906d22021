"""Image files for the render loop (extension: the reference writes CSV polylines only, cli.rs:17-30).

PFM keeps the framebuffer exactly (f32, little-endian); PPM is an 8-bit preview.  The framebuffer is cell-major,
cell = i * count_v + j (scene_config.rs:74-75); image column = i, image row = j counted from the bottom, which is PFM's
native row order.  The native binary (`raytracing --render`) writes the same PFM bytes.
"""
import numpy as np


def cells_to_rows(fb, count_u, count_v):
    """[count_u*count_v, 3] cell-major -> [count_v, count_u, 3], row j bottom-up."""
    fb = np.asarray(fb, np.float32).reshape(count_u, count_v, 3)
    return np.ascontiguousarray(fb.transpose(1, 0, 2))


def write_pfm(path, fb, count_u, count_v):
    rows = cells_to_rows(fb, count_u, count_v).astype("<f4")
    with open(path, "wb") as f:
        f.write(f"PF\n{count_u} {count_v}\n-1.0\n".encode("ascii"))
        f.write(rows.tobytes())


def read_pfm(path):
    """-> (fb cell-major [count_u*count_v, 3] f32, count_u, count_v)"""
    with open(path, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError(f"{path}: not a colour PFM")
        count_u, count_v = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4")
    if data.size != count_u * count_v * 3:
        raise ValueError(f"{path}: truncated")
    rows = data.astype(np.float32).reshape(count_v, count_u, 3)
    return np.ascontiguousarray(rows.transpose(1, 0, 2)).reshape(-1, 3), count_u, count_v


def write_ppm(path, fb, count_u, count_v, gamma=2.2):
    """8-bit preview, top row first, clamp to [0, 1] then gamma."""
    rows = cells_to_rows(fb, count_u, count_v)[::-1]
    q = np.clip(np.nan_to_num(rows, nan=0.0), 0.0, 1.0) ** (1.0 / gamma)
    with open(path, "wb") as f:
        f.write(f"P6\n{count_u} {count_v}\n255\n".encode("ascii"))
        f.write(np.rint(q * 255.0).astype(np.uint8).tobytes())
