"""Image files in and out with nothing heavier than numpy, zlib and PIL/OpenCV (no torch, no engine): decoding of the
scene's colour and depth images (loader.py:156-170) and PNG encoding of the result pictures (sucre.py:116-121), usable
in-process and from the light-weight worker processes the CLI starts (``WorkerPool``)."""
import os
import struct
import zlib

import numpy as np


def imread_rgb_u8(path) -> np.ndarray:
    """(H,W,3) uint8 RGB as stored in the file: ``cv2.cvtColor(cv2.imread(path), COLOR_BGR2RGB)`` (loader.py:157), PIL when
    OpenCV is not installed."""
    try:
        import cv2
        bgr = cv2.imread(str(path))
        if bgr is None:
            raise FileNotFoundError(path)
        return cv2.cvtColor(bgr, cv2.COLOR_BGR2RGB)
    except ImportError:
        from PIL import Image as PILImage
        with PILImage.open(path) as im:
            return np.asarray(im.convert('RGB'))


def imread_depth_u16(path) -> np.ndarray:
    """The depth image exactly as stored: ``cv2.imread(path, IMREAD_UNCHANGED)`` (loader.py:167), PIL without OpenCV."""
    try:
        import cv2
        d = cv2.imread(str(path), cv2.IMREAD_UNCHANGED)
        if d is None:
            raise FileNotFoundError(path)
        return d
    except ImportError:
        from PIL import Image as PILImage
        with PILImage.open(path) as im:
            return np.asarray(im)


def encode_rgb(px: np.ndarray, level: int = 1) -> bytes:
    """The PNG file holding exactly the (H, W, 3) uint8 pixels ``px``: Sub-filtered rows, one zlib stream.  At the speed
    setting (level <= 1) Huffman coding only: after the Sub filter string matching finds little, and on the synthetic
    survey images it is both 1.7x faster and 20 % smaller than level-1 deflate with matching."""
    px = np.ascontiguousarray(px, dtype=np.uint8)
    H, W, _ = px.shape
    rows = np.empty((H, 1 + 3 * W), np.uint8)
    rows[:, 0] = 1                                   # filter type 1 (Sub): every byte minus the same channel one pixel left
    flat = px.reshape(H, 3 * W)
    rows[:, 1:4] = flat[:, :3]
    np.subtract(flat[:, 3:], flat[:, :-3], out=rows[:, 4:])
    deflate = zlib.compressobj(level, zlib.DEFLATED, 15, 9, zlib.Z_HUFFMAN_ONLY if level <= 1 else zlib.Z_DEFAULT_STRATEGY)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2, 0, 0, 0))
            + chunk(b'IDAT', deflate.compress(rows.tobytes()) + deflate.flush()) + chunk(b'IEND', b''))


def write_rgb(path: str, px: np.ndarray, level: int = 1) -> None:
    with open(path, 'wb') as f:
        f.write(encode_rgb(px, level))


def default_level() -> int:
    return int(os.environ.get('SUCRE_PNG_COMPRESS_LEVEL', '1'))


# ---- worker processes -----------------------------------------------------------------------------------------------
# Measured on the MI355X box (tools/cli_timeline.py, 48-image 1080p survey): with the PNGs encoded by threads of the
# process that drives the GPU, the pipeline delivers an image every 35-40 ms; with the same encoding done by other
# processes, every 24 ms -- the GPU-bound rate.  zlib releases the GIL, the threads stayed within the CPU quota, and an
# idle sleep of the same length in their place cost nothing, so it is CPU work inside the GPU-driving process as such
# that slows its launches down.  Decoding has the mirror problem: sixteen PIL decode threads reach the throughput of
# four (the decoder takes the GIL for every 64 KiB block).  Hence child processes, which import neither torch nor the
# engine, for both.

_HEADER = struct.Struct('<cIIII')   # op, path bytes, H, W, zlib level   (ops: W = write RGB PNG, R = read colour, D = read depth)
_REPLY = struct.Struct('<IIIII')    # error bytes, payload bytes, H, W, channels (0 = 2-D array)

POOL = None   # the process-wide WorkerPool while the CLI runs a survey (start_pool / stop_pool)


def _worker_main() -> None:
    import sys
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        head = inp.read(_HEADER.size)
        if len(head) < _HEADER.size:
            return
        op, n_path, H, W, level = _HEADER.unpack(head)
        path = inp.read(n_path).decode()
        px = np.frombuffer(inp.read(H * W * 3), np.uint8).reshape(H, W, 3) if op == b'W' else None
        payload, shape, msg = b'', (0, 0, 0), b''
        try:
            if op == b'W':
                write_rgb(path, px, level)
            else:
                arr = np.ascontiguousarray(imread_rgb_u8(path) if op == b'R' else imread_depth_u16(path))
                shape = (arr.shape[0], arr.shape[1], arr.shape[2] if arr.ndim == 3 else 0)
                payload = arr.dtype.str.encode().ljust(8) + arr.tobytes()
        except Exception as e:   # reported to the parent, which raises
            msg = (type(e).__name__ + ': ' + str(e)).encode()
        out.write(_REPLY.pack(len(msg), len(payload), *shape) + msg + payload)
        out.flush()


class WorkerLost(RuntimeError):
    """A worker process died mid-request; the request was not served (callers may do the work themselves)."""


class WorkerPool:
    """Up to ``n`` child processes that decode image files and encode + write PNG files.  A call blocks the calling
    thread (not the GIL) until its file is read / on disk, so callers keep the semantics of the in-process functions.
    Workers are started on demand."""

    def __init__(self, n: int):
        import queue
        import threading
        self._n = max(1, n)
        self._procs = []
        self._free = queue.Queue()
        self._lock = threading.Lock()

    def _spawn(self):
        import subprocess
        import sys
        from pathlib import Path
        env = dict(os.environ, PYTHONPATH=str(Path(__file__).resolve().parent.parent) + os.pathsep + os.environ.get('PYTHONPATH', ''),
                   # a worker is one thread of work: keep numpy's BLAS / OpenMP from starting a thread per CPU of the machine
                   OPENBLAS_NUM_THREADS='1', OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
        return subprocess.Popen([sys.executable, '-c', 'from sucre_amd import _pixelio; _pixelio._worker_main()'],
                                stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)

    def _acquire(self):
        import queue
        while True:
            try:
                return self._free.get_nowait()
            except queue.Empty:
                pass
            with self._lock:
                # workers that died while idle or while serving someone else leave room for a replacement; a waiter
                # re-checks this every so often, so it cannot sleep for ever on a pool whose workers were all killed
                self._procs = [q for q in self._procs if q.poll() is None]
                if len(self._procs) < self._n:
                    p = self._spawn()
                    self._procs.append(p)
                    return p
            try:
                return self._free.get(timeout=0.5)
            except queue.Empty:
                continue

    def _retire(self, p, why: str):
        with self._lock:
            if p in self._procs:
                self._procs.remove(p)
        p.kill()
        return WorkerLost(f'image worker process died ({why})')

    def _call(self, op: bytes, path, px=None, level: int = 0):
        name = str(path).encode()
        H, W = (px.shape[0], px.shape[1]) if px is not None else (0, 0)
        p = self._acquire()
        if p.poll() is not None:      # died while idle in the free queue
            raise self._retire(p, f'exit code {p.returncode}')
        try:
            p.stdin.write(_HEADER.pack(op, len(name), H, W, level) + name)
            if px is not None:
                p.stdin.write(memoryview(px).cast('B'))
            p.stdin.flush()
            n_msg, n_payload, h, w, c = _REPLY.unpack(p.stdout.read(_REPLY.size))
            msg = p.stdout.read(n_msg)
            payload = p.stdout.read(n_payload)
        except (OSError, struct.error) as e:   # the worker is gone: retire it (another one is started on demand)
            raise self._retire(p, repr(e)) from e
        if len(msg) != n_msg or len(payload) != n_payload:   # it died mid-reply: a short read raises nothing by itself
            raise self._retire(p, f'short reply: {len(msg)}/{n_msg} + {len(payload)}/{n_payload} bytes')
        self._free.put(p)
        if msg:
            text = msg.decode()
            raise (FileNotFoundError if text.startswith('FileNotFoundError') else RuntimeError)(f'image worker process: {text}')
        if op == b'W':
            return None
        arr = np.frombuffer(payload, np.dtype(payload[:8].decode().strip()), offset=8)
        return arr.reshape((h, w, c) if c else (h, w))

    def write(self, path, px: np.ndarray, level: int = 1) -> None:
        self._call(b'W', path, np.ascontiguousarray(px, dtype=np.uint8), level)

    def read(self, path, depth: bool = False) -> np.ndarray:
        return self._call(b'D' if depth else b'R', path)

    def close(self) -> None:
        for p in self._procs:
            try:
                p.stdin.close()
            except OSError:
                pass
        for p in self._procs:
            p.wait()
        self._procs = []


def start_pool(n: int) -> WorkerPool:
    global POOL
    if POOL is None:
        POOL = WorkerPool(n)
    return POOL


def stop_pool() -> None:
    global POOL
    pool, POOL = POOL, None
    if pool is not None:
        pool.close()
