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


def _base():
    return "/dev/shm" if os.path.isdir("/dev/shm") else os.environ.get("TMPDIR", "/tmp")


def _create(tag):
    """A fresh file of this job's own: unpredictable name, O_CREAT | O_EXCL, mode 0600 (``mkstemp``).  Nobody can have put
    a file or a link of that name there before us, and the name reaches the other ranks through the collective, not through
    a convention another user of the node could guess (round 3 used /dev/shm/alfi_gen_<uid>_<MASTER_PORT>_<config>.bin)."""
    import re
    import tempfile
    return tempfile.mkstemp(prefix="alfi_gen_%s_" % re.sub(r"[^A-Za-z0-9_.-]", "_", str(tag)), suffix=".bin", dir=_base())


def _write(obj, f):
    bufs = []
    data = pickle.dumps(obj, protocol=5, buffer_callback=bufs.append)
    raws = [b.raw() for b in bufs]
    f.write(struct.pack("<8sQQ", MAGIC, len(data), len(raws)) + b"".join(struct.pack("<Q", r.nbytes) for r in raws))
    f.write(data)
    for r in raws:
        f.write(b"\0" * ((-f.tell()) % 64))
        f.write(r)
    f.flush()
    return sum(r.nbytes for r in raws)


def dump(obj, path):
    """Write ``obj`` to a NEW file ``path`` (fails if anything of that name exists; never follows a link), mode 0600."""
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    try:
        with os.fdopen(fd, "wb") as f:
            return _write(obj, f)
    except BaseException:
        try:                                       # a partial file in /dev/shm is held in memory: do not leave it behind
            os.unlink(path)
        except OSError:
            pass
        raise


def load(path):
    """The object with its arrays as copy-on-write views of the file's pages; the mapping lives as long as the arrays do.
    The file is unpickled only if it is a regular file that belongs to this user and that nobody else can write (checked on
    the OPEN descriptor, links are not followed): pickle data from anybody else is code execution."""
    import stat
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise PermissionError("%s: not a private regular file of uid %d (owner %d, mode %o): refusing to unpickle it"
                                  % (path, os.getuid(), st.st_uid, stat.S_IMODE(st.st_mode)))
        mm = mmap.mmap(fd, 0, access=mmap.ACCESS_COPY)
    finally:
        os.close(fd)
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


def build_shared(build, rank, broadcast, barrier, tag, set_threads=None, all_threads=None, my_threads=None):
    """``build()`` runs on rank 0 only (with ``all_threads`` host threads if ``set_threads`` is given), everyone returns the
    same object.  ``broadcast(x)``: a collective over the node's ranks returning rank 0's ``x`` on every rank (it carries the
    file's name, or the news that there is none); ``barrier``: a collective, called once, after which rank 0 unlinks the
    file -- nothing stays behind in /dev/shm (held in memory) when the run ends or dies later.  If the file cannot be written
    (no /dev/shm, not enough room) the other ranks build their own copy: slower, never wrong.  If the generation itself
    fails on rank 0, every rank raises."""
    obj, err, path = None, None, None
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
        if err is None:
            try:
                fd, path = _create(tag)
                try:
                    with os.fdopen(fd, "wb") as f:
                        _write(obj, f)
                except BaseException:
                    os.unlink(path)
                    raise
            except (OSError, pickle.PicklingError, MemoryError) as e:
                path = None
                import sys
                sys.stderr.write("[alfi_amd.shared] cannot share the generated hierarchy (%s): every rank builds its own\n" % (e,))
    word = broadcast(("error", repr(err)) if err is not None else ("file", path))
    try:
        if rank != 0:
            if word[0] == "error":
                raise RuntimeError("rank 0 failed to generate the hierarchy: %s" % word[1])
            obj = None
            if word[1]:
                # (a rank that cannot map the file -- wrong owner or mode, another node's /dev/shm, a truncated file --
                # builds its own copy, as promised above, instead of raising alone after the barrier)
                try:
                    obj = load(word[1])
                except (OSError, ValueError, EOFError, pickle.UnpicklingError) as e:
                    import sys
                    sys.stderr.write("[alfi_amd.shared] rank %d cannot map %s (%s): building its own copy\n" % (rank, word[1], e))
            if obj is None:
                obj = build()
    finally:
        barrier()
        if rank == 0 and path:
            try:
                os.unlink(path)
            except OSError:
                pass
    if err is not None:
        raise err
    return obj
