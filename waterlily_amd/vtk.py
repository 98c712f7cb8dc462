"""Checkpoint / visualisation I/O: the reference's VTK extensions on top of the device fields, as an ASYNCHRONOUS snapshot.

    vtkWriter / write / close    ext/WaterLilyWriteVTKExt.jl:28-72   (ImageData .vti per snapshot + a .pvd collection)
    restart_sim                   ext/WaterLilyReadVTKExt.jl:28-45    (reload u, p; reset dt[end]; push CFL)

The reference's `write!` is synchronous: `a.flow.u |> Array` (device -> host), `components_first` on the host, encode, write.
Here `write(w, sim)` only ENQUEUES:

  stepping thread   wl_snapshot_pack on the library's stream: the local planes of every attribute are packed into a dense
                    array-of-tuples DEVICE staging slot (a ring of `ring` slots in HBM -- 288 GB buys a burst of snapshots that
                    neither PCIe nor the disk has to keep up with), one event, one queue entry;
  worker thread     waits for the event on a SIDE stream, one asynchronous D2H copy of the slot into a PINNED host buffer
                    (`host_buffers` of them), releases the slot, then writes the file straight from the pinned buffer: a VTK
                    XML ImageData file with raw appended data (what WriteVTK.jl emits, minus compression) -- no base64 string,
                    no XML tree holding the data, no second host copy.

z-slab runs write one piece file per rank (its owned planes plus one shared plane) and rank 0 adds the .pvti index: no gather,
no collective, no rank-asymmetric host work.  `close` drains the queue and writes the .pvd collection.  `restart_sim` reads
the piece(s) that overlap the rank's planes (any decomposition can restart from any other) through a memory map and unpacks
them on the device.  The reader also understands WriteVTK.jl's own encodings (appended raw / base64, vtkZLibDataCompressor,
inline binary), so snapshots written by the Julia reference can seed a run here.
"""
from __future__ import annotations

import base64
import ctypes as C
import os
import queue
import re
import struct
import threading
import time as _time
import xml.etree.ElementTree as ET
import zlib
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import sim as S

_VTK_T = {np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64"}
_NP_T = {"Float32": np.float32, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8}


def _velocity(sim):
    return sim.flow.u


def _pressure(sim):
    return sim.flow.p


def default_attrib() -> Dict[str, Callable]:
    """WriteVTKExt.jl:48-50.  An attribute function returns a device field of the simulation's layout (scalar (N...) or vector
    (N..., D)): it is packed on the device, no host array is made."""
    return {"Velocity": _velocity, "Pressure": _pressure}


def S_rank() -> int:
    from . import dist
    return dist.rank_size()[0]


def _piece_planes(sim) -> Tuple[int, int, int]:
    """(first GLOBAL plane, last GLOBAL plane, local index of the first) this rank writes: the planes it owns, rank 0 down to
    the ghost plane 0, every rank but the last one more plane (shared with its upper neighbour: VTK pieces abut on a point
    layer; the plane is a current halo copy), the last rank up to the ghost plane."""
    N = tuple(sim.flow.N)
    sl = sim.slab
    if sl is None or len(N) < 3:
        return 0, (N[2] - 1 if len(N) == 3 else 0), 0
    glo = 0 if sl.rank == 0 else sl.kz0 + sl.own_lo
    ghi = N[2] - 1 if sl.rank == sl.size - 1 else sl.kz0 + sl.own_hi + 1
    return glo, ghi, glo - sl.kz0


def _all_pieces(sim) -> List[Tuple[int, int]]:
    """the plane ranges of every rank (pure slab arithmetic: rank 0 writes the index without asking anybody)"""
    sl = sim.slab
    if sl is None:
        return [_piece_planes(sim)[:2]]
    from .dist import Slab
    out = []
    for r in range(sl.size):
        s = Slab(r, sl.size, sl.nz, sl.ring)
        out.append((0 if r == 0 else s.kz0 + s.own_lo, sim.flow.N[2] - 1 if r == sl.size - 1 else s.kz0 + s.own_hi + 1))
    return out


class _Slot:
    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.free = threading.Event()
        self.free.set()


class VTKWriter:
    """WriteVTKExt.jl:27-41.  ring: device staging slots (snapshots that may be in flight before `write` has to wait);
    host_buffers: pinned host buffers of one snapshot each; on_busy: "wait" (default: like the reference, no snapshot is
    lost) or "skip" (drop the snapshot when every slot is still in flight)."""

    def __init__(self, fname="WaterLily", attrib=None, dir="vtk_data", count=0, entries=None, ring=4, host_buffers=2,
                 on_busy="wait"):
        self.fname, self.dir_name = fname, dir
        self.output_attrib = default_attrib() if attrib is None else attrib
        self.count = count
        self.entries = [] if entries is None else entries      # (time, path)
        self.on_busy = on_busy
        self._slots = [_Slot() for _ in range(max(1, ring))]
        self._host: List[Optional[torch.Tensor]] = [None] * max(1, host_buffers)
        self._hostk = 0
        self._q: "queue.Queue" = queue.Queue()
        self._err: Optional[BaseException] = None
        self._side: Optional[torch.cuda.Stream] = None
        self._thread: Optional[threading.Thread] = None
        self.stats = {"snapshots": 0, "skipped": 0, "bytes": 0, "enqueue_s": 0.0, "wait_s": 0.0, "d2h_s": 0.0, "write_s": 0.0}
        os.makedirs(dir, exist_ok=True)

    # ---- worker thread: D2H on the side stream, then the file, both off the stepping thread
    def _start(self, device):
        if self._thread is None:
            self._side = torch.cuda.Stream(device=device)
            self._thread = threading.Thread(target=self._drain, name="wl-vtk-writer", daemon=True)
            self._thread.start()

    def _drain(self):
        torch.cuda.set_device(self._side.device)      # (the current device is a per-thread setting)
        while True:
            job = self._q.get()
            try:
                if job is None:
                    return
                if self._err is None:
                    self._one(*job)
            except BaseException as e:      # surfaced by the next write / close on the stepping thread
                self._err = e
                job[0].free.set()
            finally:
                self._q.task_done()

    def _one(self, slot, ev, nbytes, path, header, blocks, index):
        k = self._hostk
        self._hostk = (k + 1) % len(self._host)
        if self._host[k] is None or self._host[k].numel() < nbytes:
            self._host[k] = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        host = self._host[k]
        t0 = _time.perf_counter()
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            host[:nbytes].copy_(slot.buf[:nbytes], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._side)
        done.synchronize()
        slot.free.set()                       # the device slot may take the next snapshot
        t1 = _time.perf_counter()
        mv = memoryview(host.numpy())
        with open(path + ".part", "wb") as f:
            f.write(header)
            for off, n in blocks:             # [UInt64 byte count][raw data] per array, straight from the pinned buffer
                f.write(struct.pack("<Q", n))
                f.write(mv[off:off + n])
            f.write(b"\n  </AppendedData>\n</VTKFile>\n")
        os.replace(path + ".part", path)
        if index is not None:
            with open(index[0], "wb") as f:
                f.write(index[1])
        t2 = _time.perf_counter()
        self.stats["d2h_s"] += t1 - t0
        self.stats["write_s"] += t2 - t1
        self.stats["bytes"] += nbytes

    def _check(self):
        if self._err is not None:
            e, self._err = self._err, None
            raise RuntimeError(f"VTK writer thread failed: {e!r}") from e


def vtkWriter(fname="WaterLily", attrib=None, dir="vtk_data", T=np.float32, **kw) -> VTKWriter:
    return VTKWriter(fname, attrib, dir, **kw)


def _header(wext, pext, arrays) -> bytes:
    """XML up to the '_' of the appended section.  arrays: (name, vtk type, ncomp, offset)"""
    out = ['<?xml version="1.0" encoding="utf-8"?>',
           '<VTKFile type="ImageData" version="1.0" byte_order="LittleEndian" header_type="UInt64">',
           f'  <ImageData WholeExtent="{wext}" Origin="0 0 0" Spacing="1 1 1">',
           f'    <Piece Extent="{pext}">', '      <PointData>']
    for name, vt, nc, off in arrays:
        out.append(f'        <DataArray type="{vt}" Name="{name}" NumberOfComponents="{nc}" format="appended" offset="{off}"/>')
    out += ['      </PointData>', '    </Piece>', '  </ImageData>', '  <AppendedData encoding="raw">', '   _']
    return "\n".join(out).encode()


def _pvti(wext, arrays, pieces) -> bytes:
    out = ['<?xml version="1.0" encoding="utf-8"?>',
           '<VTKFile type="PImageData" version="1.0" byte_order="LittleEndian" header_type="UInt64">',
           f'  <PImageData WholeExtent="{wext}" GhostLevel="0" Origin="0 0 0" Spacing="1 1 1">', '    <PPointData>']
    for name, vt, nc, _ in arrays:
        out.append(f'      <PDataArray type="{vt}" Name="{name}" NumberOfComponents="{nc}"/>')
    out.append('    </PPointData>')
    for pext, src in pieces:
        out.append(f'    <Piece Extent="{pext}" Source="{src}"/>')
    out += ['  </PImageData>', '</VTKFile>', '']
    return "\n".join(out).encode()


def write(w: VTKWriter, sim) -> None:
    """WriteVTKExt.jl:57-66: snapshot at sim_time(sim).  Returns as soon as the pack kernels are enqueued (see the module text)."""
    w._check()
    t0 = _time.perf_counter()
    flow = sim.flow
    N = tuple(flow.N)
    D = len(N)
    L = _lib.lib()
    glo, ghi, klo = _piece_planes(sim)
    npl = ghi - glo + 1
    T = flow.T
    tsz = T.itemsize
    fields = []
    nbytes = 0
    for name, func in w.output_attrib.items():
        a = func(sim)
        if not isinstance(a, torch.Tensor) or not a.is_cuda:
            raise TypeError(f"attribute {name!r}: the function must return a device field of the simulation (e.g. sim.flow.u)")
        if tuple(a.shape[:D]) != tuple(flow.p.shape):
            raise ValueError(f"attribute {name!r}: a field of extents {tuple(flow.p.shape)} (+ components) is expected")
        nc = int(np.prod(a.shape[D:])) if a.ndim > D else 1
        nct = 1 if a.ndim == D else max(3, nc)                # vectors carry 3 components (2-D: zero third one)
        n = N[0] * N[1] * npl * nct * tsz
        fields.append((name, a, nc, nct, nbytes, n))
        nbytes += (n + 255) // 256 * 256
    slot = w._slots[w.count % len(w._slots)]
    if not slot.free.is_set():
        if w.on_busy == "skip":
            w.stats["skipped"] += 1
            w.count += 1
            return
        t1 = _time.perf_counter()
        slot.free.wait()
        w.stats["wait_s"] += _time.perf_counter() - t1
        w._check()
    slot.free.clear()
    if slot.buf is None or slot.buf.numel() < nbytes:
        # sized once: every slot of the ring at the first snapshot (an allocation synchronises the device -- none later)
        torch.cuda.synchronize()
        for s in w._slots:
            if s is slot or s.free.is_set():
                s.buf = torch.empty(nbytes, dtype=torch.uint8, device=flow.device)
    w._start(flow.device)
    for name, a, nc, nct, off, n in fields:
        g = S._grid_of(a, D)
        _lib.check(L.wl_snapshot_pack(S._WLT[T], C.byref(g), C.c_void_p(a.data_ptr()), nc, nct, klo if D == 3 else 0,
                                      klo + npl - 1 if D == 3 else 0, C.c_void_p(slot.buf.data_ptr() + off)))
    ev = torch.cuda.Event()
    ev.record(torch.cuda.default_stream(flow.device))         # the stream the library works on (its default: the device's null stream)
    # file names and XML of this snapshot (no data yet)
    sl = sim.slab
    multi = sl is not None and sl.size > 1
    wext = " ".join(f"1 {n}" for n in N) + " 1 1" * (3 - D)

    def pext(lo, hi):
        return (f"1 {N[0]} 1 {N[1]} {lo + 1} {hi + 1}" if D == 3 else f"1 {N[0]} 1 {N[1]} 1 1")
    arrays, blocks, off = [], [], 0
    for name, a, nc, nct, boff, n in fields:
        arrays.append((name, _VTK_T[np.dtype(T)], nct, off))
        blocks.append((boff, n))
        off += 8 + n
    stem = f"{w.fname}_{w.count:06d}"
    rel = f"{stem}_r{sl.rank:04d}.vti" if multi else f"{stem}.vti"
    path = os.path.join(w.dir_name, os.path.basename(rel))
    index = None
    entry = path
    if multi:
        entry = os.path.join(w.dir_name, os.path.basename(stem) + ".pvti")
        if sl.rank == 0:
            pieces = [(pext(lo, hi), f"{os.path.basename(stem)}_r{r:04d}.vti") for r, (lo, hi) in enumerate(_all_pieces(sim))]
            index = (entry, _pvti(wext, arrays, pieces))
    w._q.put((slot, ev, nbytes, path, _header(wext, pext(glo, ghi), arrays), blocks, index))
    w.entries.append((round(S.sim_time(sim), 4), entry))
    w.count += 1
    w.stats["snapshots"] += 1
    w.stats["enqueue_s"] += _time.perf_counter() - t0


def flush(w: VTKWriter) -> None:
    """wait until every enqueued snapshot is on disk"""
    w._q.join()
    w._check()


def close(w: VTKWriter) -> None:
    """WriteVTKExt.jl:72: drains the queue and writes the .pvd collection (rank 0)"""
    flush(w)
    if w._thread is not None:
        w._q.put(None)
        w._thread.join()
        w._thread = None
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


def _b64(arr: np.ndarray) -> str:
    raw = np.ascontiguousarray(arr).tobytes()
    return (base64.b64encode(struct.pack("<Q", len(raw))) + base64.b64encode(raw)).decode()


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


def _extent_of(txt: str):
    e = [int(v) for v in txt.split()]
    return tuple((e[2 * d], e[2 * d + 1]) for d in range(3))


def read_vti(path: str, with_extent: bool = False):
    """Point data of an ImageData file -> {name: array shaped (n...,) or (ncomp, n...)} over the file's PIECE extent (Fortran
    point order).  Raw appended, uncompressed data (what `write` produces) is returned as views of a memory map: nothing is
    read until it is used.  with_extent: also return ((lo,hi) x 3) of the piece, 1-based like the file."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(min(size, 1 << 16))
    k = head.find(b"<AppendedData")
    appended = enc = None
    mm_off = None
    if k >= 0:                                         # the blob after the '_' marker is not valid XML: cut it out
        us = head.find(b"_", head.find(b">", k)) + 1
        enc = "base64" if b'encoding="base64"' in head[k:us] else "raw"
        xml = head[:us - 1] + b"</AppendedData></VTKFile>"
        mm_off = us
    else:
        xml = open(path, "rb").read()
    root = ET.fromstring(xml)
    htype = {"UInt64": np.uint64, "UInt32": np.uint32}[root.get("header_type", "UInt32")]
    hs = np.dtype(htype).itemsize
    compressed = root.get("compressor") is not None
    img = root.find("ImageData")
    piece = img.find("Piece")
    ext = _extent_of(piece.get("Extent") or img.get("WholeExtent"))
    N = tuple(hi - lo + 1 for lo, hi in ext)
    if mm_off is not None and not (enc == "raw" and not compressed):
        raw = open(path, "rb").read()
        appended = raw[mm_off:raw.rfind(b"</AppendedData>")]
    out = {}
    for da in piece.find("PointData").findall("DataArray"):
        dt, nc = _NP_T[da.get("type")], int(da.get("NumberOfComponents", "1"))
        fmt = da.get("format")
        shape = ((nc,) if nc > 1 else ()) + N
        cnt = int(np.prod(shape))
        if fmt == "appended" and enc == "raw" and not compressed:
            off = mm_off + int(da.get("offset"))
            arr = np.memmap(path, dtype=dt, mode="r", offset=off + hs, shape=(cnt,))
        elif fmt == "appended":
            off = int(da.get("offset"))
            if enc == "base64":
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
                    hl = 4 * ((hs + 2) // 3)
                    data = base64.b64decode(appended[off:off + hl]) + base64.b64decode(appended[off + hl:off + hl + 4 * ((n + 2) // 3)])
                arr = _decode(data, dt, compressed, htype)
            else:
                arr = _decode(appended[off:], dt, compressed, htype)
        elif fmt == "binary":
            txt = "".join(da.text.split())
            hl = 4 * ((hs + 2) // 3)
            if compressed:
                nb = int(np.frombuffer(base64.b64decode(txt[:4 * ((3 * hs + 2) // 3)])[:hs], dtype=htype)[0])
                hlen = 4 * (((3 + nb) * hs + 2) // 3)
                arr = _decode(base64.b64decode(txt[:hlen]) + base64.b64decode(txt[hlen:]), dt, True, htype)
            else:
                arr = _decode(base64.b64decode(txt[:hl]) + base64.b64decode(txt[hl:]), dt, False, htype)
        else:
            arr = np.array(da.text.split(), dtype=dt)
        out[da.get("Name")] = arr[:cnt].reshape(shape, order="F")
    return (out, ext) if with_extent else out


def read_pieces(path: str):
    """[(piece extent, file)] of a .pvti index, or the one piece of a .vti, plus the whole extent"""
    if not path.endswith(".pvti"):
        root = ET.fromstring(re.sub(rb"<AppendedData.*", b"</VTKFile>", open(path, "rb").read(1 << 16), flags=re.S))
        img = root.find("ImageData")
        return _extent_of(img.get("WholeExtent")), [(_extent_of(img.find("Piece").get("Extent") or img.get("WholeExtent")), path)]
    img = ET.parse(path).getroot().find("PImageData")
    base = os.path.dirname(os.path.abspath(path))
    return _extent_of(img.get("WholeExtent")), [(_extent_of(p.get("Extent")), os.path.join(base, p.get("Source")))
                                               for p in img.findall("Piece")]


def restart_sim(sim, fname: str = "WaterLily.pvd", attrib=None, **kw) -> VTKWriter:
    """ReadVTKExt.jl:28-45.  Every rank reads, through a memory map, the planes of the piece(s) that overlap its own (halo
    planes included), uploads them and unpacks on the device."""
    items = read_pvd(fname)
    whole, pieces = read_pieces(items[-1][1])
    flow = sim.flow
    N = tuple(flow.N)
    D = len(N)
    got = tuple(hi - lo + 1 for lo, hi in whole)[:D]
    assert got == N, "The dimensions of the simulation do not match the dimensions of the vtk file"
    sl = getattr(flow.u, "_wl_slab", None)
    # LOCAL planes that exist in the undecomposed array, as global plane numbers (0-based)
    g0 = 0 if sl is None else max(0, sl.kz0)
    g1 = (N[2] - 1 if D == 3 else 0) if sl is None else min(N[2] - 1, sl.kz0 + sl.n2l - 1)
    kz0 = 0 if sl is None else sl.kz0
    L = _lib.lib()
    T = flow.T
    for name, field, nc in (("Pressure", flow.p, 1), ("Velocity", flow.u, D)):
        for ext, path in pieces:
            plo, phi = (ext[2][0] - 1, ext[2][1] - 1) if D == 3 else (0, 0)
            lo, hi = max(g0, plo), min(g1, phi)
            if lo > hi:
                continue
            a = read_vti(path)[name]                                    # (nct,) + piece extents, Fortran order: tuples first
            nct = a.shape[0] if a.ndim == 4 else 1
            sub = a[..., lo - plo:hi - plo + 1]                          # planes lo..hi (a view of the memory map)
            flat = np.array(np.asarray(sub).reshape(-1, order="F"), dtype=T)      # (a writable copy of the mapped planes)
            dev = torch.from_numpy(flat).to(flow.device)
            g = S._grid_of(field, D)
            _lib.check(L.wl_snapshot_unpack(S._WLT[T], C.byref(g), C.c_void_p(field.data_ptr()), nc, nct,
                                            lo - kz0 if D == 3 else 0, hi - kz0 if D == 3 else 0, C.c_void_p(dev.data_ptr())))
            torch.cuda.synchronize()                                     # (`dev` is released after the kernel that reads it)
    S.halo_exchange(flow.u, 2)
    # reset time to work with the new time step
    flow.dt[-1] = float(flow.T.type(items[-1][0] * sim.L / sim.U))
    flow.dt.append(S.CFL(flow))
    return VTKWriter(fname[:-4] if fname.endswith(".pvd") else fname, attrib, os.path.dirname(items[-1][1]) or ".",
                     count=len(items), entries=list(items), **kw)
