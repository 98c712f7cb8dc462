"""Checkpoint / visualisation I/O: the reference's VTK extensions on top of the device fields.

    vtkWriter / write / close    ext/WaterLilyWriteVTKExt.jl:28-71   (ImageData .vti per snapshot + a .pvd collection)
    restart_sim                   ext/WaterLilyReadVTKExt.jl:28-45    (reload u, p; reset dt[end]; push CFL)

Files are standard VTK XML (ParaView-readable): point data "Velocity" (components first, padded to 3) and "Pressure"
on the grid 1:N, Float32/Float64 binary (base64, uncompressed).  The reader also understands what WriteVTK.jl emits
by default (appended raw/base64 data, optionally vtkZLibDataCompressor), so snapshots written by the Julia reference
can seed a run here.  Multi-GPU: fields are gathered (owned planes of every rank) and rank 0 writes.
"""
from __future__ import annotations

import base64
import os
import struct
import xml.etree.ElementTree as ET
import zlib
from typing import Callable, Dict, Optional

import numpy as np

from . import sim as S

_VTK_T = {np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64"}
_NP_T = {"Float32": np.float32, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8}


def _velocity(sim):
    return S.gather(sim.flow.u)


def _pressure(sim):
    return S.gather(sim.flow.p)


def default_attrib() -> Dict[str, Callable]:
    """WriteVTKExt.jl:48-50"""
    return {"Velocity": _velocity, "Pressure": _pressure}


class VTKWriter:
    """WriteVTKExt.jl:28-41"""

    def __init__(self, fname="WaterLily", attrib=None, dir="vtk_data", count=0, entries=None):
        self.fname, self.dir_name = fname, dir
        self.output_attrib = default_attrib() if attrib is None else attrib
        self.count = count
        self.entries = [] if entries is None else entries      # (time, relative path)
        if S_rank() == 0:
            os.makedirs(dir, exist_ok=True)


def S_rank() -> int:
    from . import dist
    return dist.rank_size()[0]


def vtkWriter(fname="WaterLily", attrib=None, dir="vtk_data", T=np.float32) -> VTKWriter:
    return VTKWriter(fname, attrib, dir)


def _b64(arr: np.ndarray) -> str:
    raw = np.ascontiguousarray(arr).tobytes()
    return (base64.b64encode(struct.pack("<Q", len(raw))) + base64.b64encode(raw)).decode()


def write(w: VTKWriter, sim) -> None:
    """WriteVTKExt.jl:57-66: snapshot at sim_time(sim)"""
    N = tuple(sim.flow.N)
    D = len(N)
    ext = " ".join(f"1 {n}" for n in N) + " 1 1" * (3 - D)
    arrays = []
    for name, func in w.output_attrib.items():
        a = np.asarray(func(sim))                                 # every rank takes part in the gather
        if a.shape == N:
            flat, ncomp = np.asfortranarray(a).ravel(order="F"), 1
        else:                                                      # components first (WriteVTKExt.jl:79), padded to 3
            nc = a.shape[-1]
            v = np.zeros(N + (3,), dtype=a.dtype)
            v[..., :nc] = a
            flat, ncomp = np.moveaxis(v, -1, 0).ravel(order="F"), 3
        arrays.append((name, flat, ncomp))
    if S_rank() == 0:
        rel = f"{w.fname}_{w.count:06d}.vti"
        root = ET.Element("VTKFile", type="ImageData", version="1.0", byte_order="LittleEndian", header_type="UInt64")
        img = ET.SubElement(root, "ImageData", WholeExtent=ext, Origin="0 0 0", Spacing="1 1 1")
        piece = ET.SubElement(img, "Piece", Extent=ext)
        pd = ET.SubElement(piece, "PointData")
        for name, flat, ncomp in arrays:
            da = ET.SubElement(pd, "DataArray", type=_VTK_T[np.dtype(flat.dtype)], Name=name,
                               NumberOfComponents=str(ncomp), format="binary")
            da.text = _b64(flat)
        ET.ElementTree(root).write(os.path.join(w.dir_name, rel), xml_declaration=True, encoding="utf-8")
        w.entries.append((round(S.sim_time(sim), 4), os.path.join(w.dir_name, rel)))
    w.count += 1


def close(w: VTKWriter) -> None:
    """WriteVTKExt.jl:72: writes the .pvd collection"""
    if S_rank() != 0:
        return
    root = ET.Element("VTKFile", type="Collection", version="1.0", byte_order="LittleEndian")
    col = ET.SubElement(root, "Collection")
    for t, path in w.entries:
        ET.SubElement(col, "DataSet", timestep=repr(float(t)), part="0", file=path)
    ET.ElementTree(root).write(w.fname + ".pvd", xml_declaration=True, encoding="utf-8")


# ----------------------------------------------------------------------------- reading

def read_pvd(fname: str):
    col = ET.parse(fname).getroot().find("Collection")
    items = [(float(d.get("timestep")), d.get("file")) for d in col.findall("DataSet")]
    base = os.path.dirname(os.path.abspath(fname))
    return [(t, f if os.path.isabs(f) or os.path.exists(f) else os.path.join(base, f)) for t, f in items]


def _decode(buf: bytes, dtype, compressed: bool, htype) -> np.ndarray:
    hs = np.dtype(htype).itemsize
    if not compressed:
        n = int(np.frombuffer(buf[:hs], dtype=htype)[0])
        return np.frombuffer(buf[hs:hs + n], dtype=dtype)
    nb, _, _ = (int(v) for v in np.frombuffer(buf[:3 * hs], dtype=htype))
    sizes = np.frombuffer(buf[3 * hs:(3 + nb) * hs], dtype=htype)
    off, out = (3 + nb) * hs, []
    for s in sizes:
        out.append(zlib.decompress(buf[off:off + int(s)]))
        off += int(s)
    return np.frombuffer(b"".join(out), dtype=dtype)


def read_vti(path: str) -> Dict[str, np.ndarray]:
    """Point data of an ImageData file -> {name: array shaped (N...,) or (ncomp, N...)} (Fortran point order)."""
    raw = open(path, "rb").read()
    appended = None
    head = raw
    k = raw.find(b"<AppendedData")
    if k >= 0:                                         # binary blob after the '_' marker is not valid XML: cut it out
        us = raw.find(b"_", raw.find(b">", k)) + 1
        end = raw.rfind(b"</AppendedData>")
        appended, enc = raw[us:end], "base64" if b'encoding="base64"' in raw[k:us] else "raw"
        head = raw[:us - 1] + raw[end:]
    root = ET.fromstring(head)
    htype = {"UInt64": np.uint64, "UInt32": np.uint32}[root.get("header_type", "UInt32")]
    compressed = root.get("compressor") is not None
    img = root.find("ImageData")
    we = [int(v) for v in img.get("WholeExtent").split()]
    N = tuple(we[2 * d + 1] - we[2 * d] + 1 for d in range(3))
    out = {}
    for da in img.find("Piece").find("PointData").findall("DataArray"):
        dt, nc = _NP_T[da.get("type")], int(da.get("NumberOfComponents", "1"))
        fmt = da.get("format")
        if fmt == "appended":
            off = int(da.get("offset"))
            if enc == "base64":
                hs = np.dtype(htype).itemsize
                first = base64.b64decode(appended[off:off + 4 * ((3 * hs + 2) // 3) + 8])
                if compressed:
                    nb = int(np.frombuffer(first[:hs], dtype=htype)[0])
                    hlen = 4 * (((3 + nb) * hs + 2) // 3)
                    header = base64.b64decode(appended[off:off + hlen])
                    sizes = np.frombuffer(header[3 * hs:(3 + nb) * hs], dtype=htype)
                    blen = 4 * ((int(sizes.sum()) + 2) // 3)
                    data = header[:(3 + nb) * hs] + base64.b64decode(appended[off + hlen:off + hlen + blen])
                else:
                    n = int(np.frombuffer(first[:hs], dtype=htype)[0])
                    data = base64.b64decode(appended[off:off + 4 * ((hs + 2) // 3) + 4 * ((n + 2) // 3) + 8])
                    hl = 4 * ((hs + 2) // 3)
                    data = base64.b64decode(appended[off:off + hl]) + base64.b64decode(appended[off + hl:off + hl + 4 * ((n + 2) // 3)])
                arr = _decode(data, dt, compressed, htype)
            else:
                arr = _decode(appended[off:], dt, compressed, htype)
        elif fmt == "binary":
            txt = "".join(da.text.split())
            hs = np.dtype(htype).itemsize
            hl = 4 * ((hs + 2) // 3)
            if compressed:
                nb = int(np.frombuffer(base64.b64decode(txt[:4 * ((3 * hs + 2) // 3)])[:hs], dtype=htype)[0])
                hlen = 4 * (((3 + nb) * hs + 2) // 3)
                arr = _decode(base64.b64decode(txt[:hlen]) + base64.b64decode(txt[hlen:]), dt, True, htype)
            else:
                arr = _decode(base64.b64decode(txt[:hl]) + base64.b64decode(txt[hl:]), dt, False, htype)
        else:
            arr = np.array(da.text.split(), dtype=dt)
        shape = ((nc,) if nc > 1 else ()) + N
        out[da.get("Name")] = arr[: int(np.prod(shape))].reshape(shape, order="F")
    return out


def restart_sim(sim, fname: str = "WaterLily.pvd", attrib=None) -> VTKWriter:
    """ReadVTKExt.jl:28-45"""
    items = read_pvd(fname)
    data = read_vti(items[-1][1])
    N = tuple(sim.flow.N)
    D = len(N)
    p = np.squeeze(data["Pressure"])
    assert tuple(p.shape) == N, "The dimensions of the simulation do not match the dimensions of the vtk file"
    u = np.moveaxis(np.squeeze(data["Velocity"]), 0, -1)[..., :D]            # components last
    _scatter(sim.flow.p, p)
    _scatter(sim.flow.u, u)
    S.halo_exchange(sim.flow.u, 2)
    # reset time to work with the new time step
    sim.flow.dt[-1] = float(sim.flow.T.type(items[-1][0] * sim.L / sim.U))
    sim.flow.dt.append(S.CFL(sim.flow))
    return VTKWriter(fname[:-4] if fname.endswith(".pvd") else fname, attrib, os.path.dirname(items[-1][1]) or ".",
                     count=len(items), entries=list(items))


def _scatter(field, full: np.ndarray) -> None:
    """upload the undecomposed host array (every rank loads the file) into this rank's slab / field"""
    sl = getattr(field, "_wl_slab", None)
    if sl is None:
        S.upload(field, full.astype(S._T(field)))
        return
    lo = max(0, sl.kz0)
    hi = min(full.shape[2], sl.kz0 + sl.n2l)
    h = S.to_host(field)
    h[:, :, lo - sl.kz0:hi - sl.kz0] = full[:, :, lo:hi]
    S.upload(field, h)
