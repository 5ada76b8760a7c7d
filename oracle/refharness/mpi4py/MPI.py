"""One-process MPI world: every collective is the identity, self-sends are
queued and matched by the next Recv (the reference's slab<->domain remaps
Isend/Recv to self)."""
import collections
import numpy as np

SUM, MAX, MIN, LOR, LAND, PROD = 'SUM', 'MAX', 'MIN', 'LOR', 'LAND', 'PROD'
IN_PLACE = 'IN_PLACE'
ANY_SOURCE = -1
ANY_TAG = -1


def Get_processor_name():
    return 'localhost'


def _unwrap(buf):
    # upper-case calls get (buffer, dtype_char) or (buffer, sizes[, displs], ...) tuples
    if isinstance(buf, (tuple, list)) and len(buf) >= 1 and not np.isscalar(buf[0]):
        return buf[0]
    return buf


def _copy(send, recv):
    send = _unwrap(send)
    recv = _unwrap(recv)
    if send is IN_PLACE or recv is None:
        return
    s = np.asarray(send)
    r = np.asarray(recv)
    if r.size == 0:
        return
    r.reshape(-1)[:s.size] = s.reshape(-1)


class _Request:
    def wait(self, *a, **k):
        return None
    Wait = wait

    def test(self, *a, **k):
        return True, None


class _Status:
    def Get_source(self):
        return 0

    def Get_tag(self):
        return 0


Status = _Status


class _Comm:
    size = 1
    rank = 0

    def __init__(self):
        self._queue = collections.deque()
        self._oqueue = collections.deque()

    def Get_size(self):
        return 1

    def Get_rank(self):
        return 0

    def Barrier(self):
        pass

    barrier = Barrier

    def Abort(self, errorcode=1):
        # An ordinary Exception (not SystemExit): the reference wraps open() in
        # an abort-on-error decorator and then catches Exception around
        # optional CLASS header reads (linear.py:3729-3737).
        raise RuntimeError(f'MPI stub Abort({errorcode})')

    # pickled / object collectives
    def bcast(self, obj=None, root=0):
        return obj

    def allgather(self, obj):
        return [obj]

    def gather(self, obj, root=0):
        return [obj]

    def allreduce(self, obj, op=SUM):
        return obj

    def reduce(self, obj, op=SUM, root=0):
        return obj

    def sendrecv(self, sendobj, dest=0, sendtag=0, recvbuf=None, source=0, recvtag=0, status=None):
        return sendobj

    def iprobe(self, source=ANY_SOURCE, tag=ANY_TAG, status=None):
        return bool(self._oqueue)

    def isend(self, obj, dest=0, tag=0):
        self._oqueue.append(obj)
        return _Request()

    def send(self, obj, dest=0, tag=0):
        self._oqueue.append(obj)

    def recv(self, buf=None, source=ANY_SOURCE, tag=ANY_TAG, status=None):
        return self._oqueue.popleft()

    # buffer collectives
    def Allreduce(self, sendbuf, recvbuf, op=SUM):
        _copy(sendbuf, recvbuf)

    def Reduce(self, sendbuf, recvbuf, op=SUM, root=0):
        _copy(sendbuf, recvbuf)

    def Allgather(self, sendbuf, recvbuf):
        _copy(sendbuf, recvbuf)

    Allgatherv = Allgather

    def Gather(self, sendbuf, recvbuf, root=0):
        _copy(sendbuf, recvbuf)

    Gatherv = Gather

    def Bcast(self, buf, root=0):
        pass

    def Sendrecv(self, sendbuf, dest=0, sendtag=0, recvbuf=None, source=ANY_SOURCE,
                 recvtag=ANY_TAG, status=None):
        _copy(sendbuf, recvbuf)

    def Isend(self, buf, dest=0, tag=0):
        self._queue.append(np.array(np.asarray(_unwrap(buf)), copy=True))
        return _Request()

    def Send(self, buf, dest=0, tag=0):
        self._queue.append(np.array(np.asarray(_unwrap(buf)), copy=True))

    def Recv(self, buf, source=ANY_SOURCE, tag=ANY_TAG, status=None):
        _copy(self._queue.popleft(), buf)


COMM_WORLD = _Comm()
