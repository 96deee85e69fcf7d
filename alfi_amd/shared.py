"""Generate once, map everywhere: the integer side of a hierarchy (meshes, numbering, node graphs, patches, transfer stencils
-- what every rank of a partitioned run needs in full, because the partitioner works on global indices) is built by ONE
process of the node with all of the job's host threads and handed to the other ranks through a file in /dev/shm that they
map copy-on-write.  Against every rank generating its own copy: 1/N of the host CPU time, one resident copy of the arrays
instead of N, and the generating rank is not slowed down by N - 1 others competing for the memory bandwidth.

The reference distributes the mesh itself (DMPlex partition + overlap, alfi/solver.py:604-605, 661-662); here the host
generator plays Firedrake's role on one node, so sharing its output between the node's ranks is the equivalent step.

Format: pickle protocol 5; every contiguous NumPy array travels out of band, 64-byte aligned behind the pickle stream, and
is reconstructed as a view of the mapping (no copy; pages are shared between the ranks until somebody writes)."""
import mmap
import os
import pickle
import struct

MAGIC = b"ALFISHM1"


def _path(tag):
    base = "/dev/shm" if os.path.isdir("/dev/shm") else os.environ.get("TMPDIR", "/tmp")
    return os.path.join(base, "alfi_gen_%d_%s.bin" % (os.getuid(), tag))


def dump(obj, path):
    bufs = []
    data = pickle.dumps(obj, protocol=5, buffer_callback=bufs.append)
    raws = [b.raw() for b in bufs]
    header = struct.pack("<8sQQ", MAGIC, len(data), len(raws)) + b"".join(struct.pack("<Q", r.nbytes) for r in raws)
    tmp = path + ".tmp.%d" % os.getpid()
    try:
        with open(tmp, "wb") as f:
            f.write(header)
            f.write(data)
            for r in raws:
                pad = (-f.tell()) % 64
                f.write(b"\0" * pad)
                f.write(r)
        os.replace(tmp, path)
    except BaseException:
        try:                                       # a partial file in /dev/shm is held in memory: do not leave it behind
            os.unlink(tmp)
        except OSError:
            pass
        raise
    return sum(r.nbytes for r in raws)


def load(path):
    """The object with its arrays as copy-on-write views of the file's pages; the mapping lives as long as the arrays do."""
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)
    magic, ndata, nbuf = struct.unpack_from("<8sQQ", mm, 0)
    if magic != MAGIC:
        raise ValueError("%s is not a shared-hierarchy file" % path)
    off = 24
    sizes = struct.unpack_from("<%dQ" % nbuf, mm, off)
    off += 8 * nbuf
    view = memoryview(mm)
    data = view[off:off + ndata]
    off += ndata
    buffers = []
    for n in sizes:
        off += (-off) % 64
        buffers.append(view[off:off + n])
        off += n
    return pickle.loads(data, buffers=buffers)


def build_shared(build, rank, barrier, tag, set_threads=None, all_threads=None, my_threads=None):
    """``build()`` runs on rank 0 only (with ``all_threads`` host threads if ``set_threads`` is given), everyone returns the
    same object.  ``barrier``: a collective over the node's ranks (called twice).  The file is unlinked as soon as every rank
    has mapped it, so nothing stays behind in /dev/shm (held in memory) when the run ends or dies later.  If the file cannot
    be written (no /dev/shm, not enough room) the other ranks fall back to building their own copy: slower, never wrong."""
    path = _path(tag)
    obj, err = None, None
    if rank == 0:
        try:
            if set_threads and all_threads:
                set_threads(all_threads)
            obj = build()
        except BaseException as e:                 # the others must not wait for a file that never comes
            err = e
        finally:
            if set_threads and my_threads:
                set_threads(my_threads)
        try:
            if err is None:
                dump(obj, path)
            else:
                raise OSError("generation failed")
        except (OSError, pickle.PicklingError, MemoryError) as e:
            try:
                with open(path, "wb") as f:        # a marker the others recognise (bad magic)
                    f.write(b"UNAVAILABLE")
            except OSError:
                pass
            if err is None:
                import sys
                sys.stderr.write("[alfi_amd.shared] cannot share the generated hierarchy (%s): every rank builds its own\n" % (e,))
    barrier()
    try:
        if rank != 0:
            try:
                obj = load(path)
            except (OSError, ValueError, struct.error, pickle.UnpicklingError):
                obj = build()                      # rank 0 could not write the file (or failed: then this fails the same way)
    finally:
        barrier()
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    if err is not None:
        raise err
    return obj
