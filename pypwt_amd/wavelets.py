"""Python `Wavelets` class: drop-in for pycudwt / pypwt's Cython class.

Mirrors the reference class member for member (reference src/pypwt.pyx:64-615): same
constructor signature, read-only attributes, `coeffs` layout `[A, [H1, V1, D1], ...]`
(2D) / `[A, D1, ...]` (1D, each of shape (Nr, Nc_l)), level clamping, state rules, and
error behaviour -- but every method calls the gfx950 HIP library through the C ABI
(include/pypwt_amd.h) instead of binding the C++ class.

Deliberate differences from the reference (documented in DESIGN.md):
  * unknown wavelet name -> ValueError (the reference hangs in w_ilog2)
  * HIP errors raise instead of being printed and ignored
  * filter banks are per instance (the reference shares one __constant__ bank per process)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PdwtInfo, check, handle_t


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _device_array(obj, dtype):
    """(device pointer, shape) of an object that lives in GPU memory -- anything exposing
    ``__cuda_array_interface__`` (PyTorch-ROCm tensors, CuPy-ROCm arrays, the DeviceArray views below) --
    or None for host data.  Only C-contiguous arrays of the instance's dtype are accepted: the library copies
    device-to-device, it does not convert."""
    iface = getattr(obj, "__cuda_array_interface__", None)
    if iface is None:
        return None
    want = np.dtype(dtype)
    if np.dtype(iface["typestr"]) != want:
        raise ValueError("device array has dtype %s, expected %s" % (iface["typestr"], want.str))
    shape = tuple(int(x) for x in iface["shape"])
    strides = iface.get("strides")
    if strides is not None:
        expect, acc = [], want.itemsize
        for n in reversed(shape):
            expect.append(acc)
            acc *= n
        if tuple(strides) != tuple(reversed(expect)):
            raise ValueError("device array must be C-contiguous")
    # CUDA Array Interface v3 consumer rule: the producer's stream, on which the data is (or will be) valid.
    #   key absent (v2 producers, e.g. torch tensors): unknown -> the consumer synchronises the device
    #   None: no synchronisation needed;  1: legacy default stream;  2: per-thread default stream;  else: a handle
    if "stream" not in iface:
        order = ("device",)
    elif iface["stream"] is None:
        order = ()
    else:
        st = int(iface["stream"])
        if st == 0:
            raise ValueError("__cuda_array_interface__: stream 0 is not allowed (1 = legacy default, 2 = per-thread default)")
        order = ("stream", {1: 0, 2: 2}.get(st, st))  # hipStreamLegacy = 0 (NULL), hipStreamPerThread = 2
    return int(iface["data"][0]), shape, order


def _read_norms(lib, h, view):
    """(sum |c|, sum c^2) out of a two-double device slot: a blocking device-to-host pdwt_copy on the plan's stream."""
    if view is None:
        ptr = C.c_void_p()
        check(lib.pdwt_norms_slot(h, C.byref(ptr)), "pdwt_norms_slot", lib)
        addr = ptr.value
    else:
        addr = int(view.__cuda_array_interface__["data"][0])
    host = np.zeros(2, dtype=np.float64)
    count = 16 // C.sizeof(lib.pdwt_real)  # pdwt_copy counts elements of pdwt_real
    check(lib.pdwt_copy(h, host.ctypes.data_as(C.c_void_p), C.c_void_p(addr), count, 2), "pdwt_copy", lib)
    return float(host[0]), float(host[1])


class _Shape(object):
    def __init__(self, shape):
        self.shape = tuple(shape)
        self.ndim = len(self.shape)


class DeviceArray(object):
    """A borrowed, zero-copy view of a plan-owned device buffer (the image or one sub-band): exposes
    ``__cuda_array_interface__`` (version 3, with the plan's stream), so ``torch.as_tensor(view, device="cuda")``
    or ``cupy.asarray(view)`` wrap it without a copy.  Valid while the owning ``Wavelets`` lives
    (reference: image_int_ptr / coeff_int_ptr hand out raw addresses, src/pypwt.pyx:578-592)."""

    def __init__(self, owner, ptr, shape, dtype, stream):
        self._owner = owner  # keeps the plan alive
        self.ptr = int(ptr)
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False),
                                         "version": 3, "strides": None, "stream": int(stream) if stream else None}

    @property
    def nbytes(self):
        return int(np.prod(self.shape)) * self.dtype.itemsize


class Wavelets(object):
    """
    A wavelet-transform plan bound to one image (or one batch of rows) on the GPU: the drop-in for pycudwt's class of the
    same name (constructor signature, attributes and coefficient layout of src/pypwt.pyx:64-615).

    Arguments, in the reference's order:

    * ``img``      the data, float32: a 2D array (one image), a 1D array (one signal), or a 2D array with ``ndim=1``
      (every row is a signal of its own).  A device array (``__cuda_array_interface__``, a torch tensor on the GPU,
      a ``DeviceArray`` of another plan) is taken over without a trip through the host.
    * ``wname``    one of the 72 built-in wavelet names ("haar", "db2" ... "db20", "sym2" ..., "coif1" ..., "bior1.3" ...,
      "rbio1.3" ...); custom banks go in afterwards through ``set_wavelets_filters``.
    * ``levels``   how many decomposition levels are wanted; clamped to what the image size allows, with the reference's
      warning on stdout.
    * ``do_separable``       nonzero (default): row and column passes; zero: the 2D non-separable kernels.
    * ``do_cycle_spinning``  nonzero: every ``forward`` circularly shifts the image by a random offset that the matching
      ``inverse`` undoes -- the usual trick against blocking artefacts in iterative shrinkage.
    * ``do_swt``   nonzero: the undecimated (a-trous) transform, every band at full size.
    * ``ndim``     2 or 1, see ``img``.
    * ``copy``     accepted and ignored like the reference's (its copy branch is commented out, pypwt.pyx:121-142).
    """

    # storage / arithmetic type: float32 like the reference's Python class; the Wavelets64 subclass below
    # binds the fp64 build of the library (the reference's DOUBLEPRECISION compile-time variant)
    _dtype = np.float32
    _variant = "f32"

    def _check(self, rc, what=""):
        return check(rc, what, self._lib)

    def _fptr(self, a):
        return a.ctypes.data_as(C.POINTER(self._lib.pdwt_real))

    def __init__(self, img, wname, levels, do_separable=1, do_cycle_spinning=0, do_swt=0, ndim=2, copy=None):
        self._h = None
        self._lib = _lib.load(self._variant)
        # NEW: an image that already lives on the GPU is taken over device-to-device
        # (pdwt_create with mem_is_on_host = 0, pdwt/src/wt.cu:117-126); `dev` = (pointer, shape) or None
        dev = _device_array(img, self._dtype)
        if dev is None:
            img = self._checkarray(np.asarray(img))
        else:
            img = _Shape(dev[1])  # shape carrier only: nothing is copied to the host

        ndim = min(int(ndim), 2)  # src/pypwt.pyx:145
        self.batched1d = 0
        if img.ndim == 2:
            self.Nr, self.Nc = int(img.shape[0]), int(img.shape[1])
            if img.ndim != ndim:
                self.batched1d = 1
        elif img.ndim == 1:  # 1D: Nr = 1, Nc = len   (src/pypwt.pyx:152-154)
            self.Nr, self.Nc = 1, int(img.shape[0])
        else:
            raise NotImplementedError("Wavelets(): Only 1D and 2D transforms are supported for now")
        self.shape = tuple(int(s) for s in img.shape)
        self.wname = str(wname)
        self._wname = self.wname.encode("ASCII")
        self.levels = int(levels)
        self.do_separable = int(do_separable)
        self.do_cycle_spinning = int(do_cycle_spinning)
        self.do_swt = int(do_swt)
        self.ndim = img.ndim

        h = handle_t()
        if dev is not None and dev[2]:  # no plan (and no plan stream) yet: wait on the host for the image's producer,
            # on the device that OWNS the array (not necessarily the current one, where the plan is about to be built)
            self._check(self._lib.pdwt_sync_producer(self._owner_device(dev[0]),
                                                     C.c_void_p(dev[2][1] if dev[2][0] == "stream" else 0),
                                                     1 if dev[2][0] == "device" else 0), "Wavelets()")
        src = self._fptr(img) if dev is None else C.cast(C.c_void_p(dev[0]), C.POINTER(self._lib.pdwt_real))
        rc = self._lib.pdwt_create(src, self.Nr, self.Nc, self._wname, self.levels, 1 if dev is None else 0,
                                   self.do_separable, self.do_cycle_spinning, self.do_swt, ndim, C.byref(h))
        self._check(rc, "Wavelets()")
        self._h = h
        # read back what the library clamped (src/pypwt.pyx:181-183)
        info, sep, cyc, st, b = PdwtInfo(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.pdwt_get_info(self._h, C.byref(info), C.byref(sep), C.byref(cyc), C.byref(st), C.byref(b)))
        self.levels = int(info.nlevels)
        self.hlen = int(info.hlen)
        self.do_separable = int(sep.value)
        self.sizes = self._compute_sizes()

        # host-side coefficient list (src/pypwt.pyx:187-205).  The arrays are views of ONE host buffer laid out like the
        # plan's coefficient region (bands back to back in `num` order, each padded to 64 elements), so that `coeffs` is a
        # single device-to-host copy instead of one blocking copy per band (28 for a 512^2 haar transform)
        two_d = not ((self.ndim < 2) or self.batched1d)
        nb = 1 + (3 if two_d else 1) * self.levels
        offs = (C.c_longlong * nb)()
        region = int(self._lib.pdwt_coeff_region(self._h, offs, nb))
        self._cbuf = np.zeros(max(region, 1), dtype=self._dtype)

        def view(num, shape):
            n = int(shape[0]) * int(shape[1])
            return self._cbuf[int(offs[num]):int(offs[num]) + n].reshape(shape)

        self._coeffs = [view(0, self.sizes[-1])]
        for i in range(self.levels):
            if two_d:
                self._coeffs.append([view(1 + 3 * i + k, self.sizes[i]) for k in range(3)])
            else:
                self._coeffs.append(view(1 + i, self.sizes[i]))

    # -- reference: info / __repr__ / __str__ (src/pypwt.pyx:209-221)
    def info(self):
        """Print some information on the current ``Wavelets`` instance."""
        buf = C.create_string_buffer(2048)
        self._check(self._lib.pdwt_info_string(self._h, buf, len(buf)))
        print(buf.value.decode("utf-8", "replace"), end="")

    def __repr__(self):
        self.info()
        return ""

    def __str__(self):
        self.info()
        return ""

    @classmethod
    def _checkarray(cls, arr, shp=None):  # src/pypwt.pyx:224-235
        res = arr
        if arr.dtype != cls._dtype or not arr.flags["C_CONTIGUOUS"]:
            res = np.ascontiguousarray(arr, dtype=cls._dtype)
        if shp is not None:
            if arr.ndim != len(shp):
                raise ValueError("Invalid number of dimensions (expected %d, got %d)" % (len(shp), arr.ndim))
            for i in range(arr.ndim):
                if arr.shape[i] != shp[i]:
                    raise ValueError("The image does not have the correct shape (expected %s, got %s)"
                                     % (str(shp), str(arr.shape)))
        return res

    @staticmethod
    def div2(n):
        """Returns (N + (N%2))/2: the image size at the next scale."""
        return (n + (n & 1)) // 2

    def _compute_sizes(self):  # src/pypwt.pyx:247-258
        Nr, Nc = self.Nr, self.Nc
        if self.do_swt:
            return [(Nr, Nc)] * self.levels
        res = []
        for _ in range(self.levels):
            Nc = self.div2(Nc)
            if not self.batched1d:
                Nr = self.div2(Nr)
            res.append((Nr, Nc))
        return res

    def coeff_only(self, num):
        """
        Get only the coeff "num" from the device.

        num : int
            2D : [0: A, 1: H1, 2: V1, 3: D1,  4: H2, ...]
            1D : [0: A, 1: D1, 2: D2, ...]
        """
        num = int(num)
        if num == 0:
            coeff_ref = self._coeffs[0]
        elif (self.ndim == 2) and not self.batched1d:
            coeff_ref = self._coeffs[(num - 1) // 3 + 1][(num - 1) % 3]
        else:
            coeff_ref = self._coeffs[num]
        numc = self._lib.pdwt_get_coeff(self._h, _ptr(coeff_ref), num)
        if numc != coeff_ref.size:  # src/pypwt.pyx:284-285 (0 when refused after inverse())
            raise RuntimeError("Wavelets.coeff_only(): something went wrong when retrieving coefficients numbef %d, "
                               "expected %d coeffs, got %d (%s)" % (num, coeff_ref.size, numc, _lib.last_error(self._lib)))
        return coeff_ref

    @property
    def coeffs(self):
        """
        Get all the coefficients from the device.
        Returns the list [A, [H1, V1, D1], [H2, V2, D2], ...] (2D) or [A, D1, ...] (1D).
        """
        numc = self._lib.pdwt_get_coeff_region(self._h, _ptr(self._cbuf))  # ONE copy: the arrays are views of _cbuf
        if numc != self._cbuf.size:  # src/pypwt.pyx:284-285 (0 when refused after inverse())
            raise RuntimeError("Wavelets.coeffs: something went wrong when retrieving coefficients, expected %d coeffs, "
                               "got %d (%s)" % (self._cbuf.size, numc, _lib.last_error(self._lib)))
        return self._coeffs

    @property
    def image(self):
        res = np.zeros((self.Nr, self.Nc), dtype=self._dtype)
        numc = self._lib.pdwt_get_image(self._h, _ptr(res))
        if numc != res.size:
            raise RuntimeError("Wavelets.image(): something went wrong when retrieving image, expected %d coeffs, "
                               "got %d (%s)" % (res.size, numc, _lib.last_error(self._lib)))
        return res

    def _owner_device(self, ptr):
        """Device ordinal that owns a device allocation, -1 (= the current device) when the runtime cannot tell."""
        d = int(self._lib.pdwt_device_of_pointer(C.c_void_p(ptr)))
        return d if d >= 0 else -1

    def _order_after_producer(self, dev):
        """The copy of a device array runs on the plan's own (non-blocking) stream: order it after the array's
        producer first (an event wait when the producer's stream is known, a device synchronisation when it is not)."""
        if not dev[2]:
            return
        if dev[2][0] == "stream":
            self._check(self._lib.pdwt_wait_for_stream(self._h, C.c_void_p(dev[2][1])))
        else:  # producer stream unknown (CAI v2): synchronise the device that owns the array -- the producer ran there
            owner = self._owner_device(dev[0])
            self._check(self._lib.pdwt_sync_producer(owner if owner >= 0 else self._lib.pdwt_device(self._h), None, 1))

    def _set_image_any(self, img, shp):
        dev = _device_array(img, self._dtype)
        if dev is not None:  # device-to-device (pdwt_set_image with mem_is_on_device = 1, wt.cu:425-431)
            if int(np.prod(dev[1])) != self.Nr * self.Nc:
                raise ValueError("The image does not have the correct shape (expected %s, got %s)" % (str(shp), str(dev[1])))
            self._order_after_producer(dev)
            self._check(self._lib.pdwt_set_image(self._h, C.c_void_p(dev[0]), 1))
            return
        img = self._checkarray(np.asarray(img), shp)
        self._check(self._lib.pdwt_set_image(self._h, _ptr(img), 0))

    def set_image(self, img):
        """Replace the image (does not update the coefficients; run forward()).  Host arrays are uploaded;
        device arrays (``__cuda_array_interface__``) are copied device-to-device on the plan's stream."""
        self._set_image_any(img, (self.Nr, self.Nc))

    def forward(self, img=None):
        """Forward wavelet transform of ``img`` if given, else of the current image."""
        if img is not None:
            self._set_image_any(img, self.shape)
        self._check(self._lib.pdwt_forward(self._h), "forward")

    def inverse(self):
        """
        Inverse transform: coefficients -> ``Wavelets.image``.

        As in the reference, calling it twice in a row does nothing but warn, and the
        coefficients cannot be read or thresholded afterwards until forward() is run again.
        """
        rc = self._lib.pdwt_inverse(self._h)
        if rc == _lib.ERR_STATE:  # reference: puts() a warning and returns (wt.cu:272-279)
            print("Warning: " + _lib.last_error(self._lib))
            return
        self._check(rc, "inverse")

    def _threshold(self, fn, beta, do_threshold_appcoeffs, normalize):
        rc = fn(self._h, float(beta), int(do_threshold_appcoeffs), int(normalize))
        if rc == _lib.ERR_STATE:  # wt.cu:309-312
            print("Warning: Wavelets(): " + _lib.last_error(self._lib))
            return
        self._check(rc)

    def soft_threshold(self, beta, do_threshold_appcoeffs=0, normalize=0):
        """ST(x, t) = (|x| - t)_+ . sign(x) on the detail (optionally approximation) coefficients;
        ``normalize``: t is divided by sqrt(2) at each scale."""
        self._threshold(self._lib.pdwt_soft_threshold, beta, do_threshold_appcoeffs, normalize)

    def hard_threshold(self, beta, do_threshold_appcoeffs=0, normalize=0):
        """HT(x, t) = x . 1_{|x| > t}"""
        self._threshold(self._lib.pdwt_hard_threshold, beta, do_threshold_appcoeffs, normalize)

    def group_soft_threshold(self, beta, do_threshold_appcoeffs=0, normalize=0):
        """Per-pixel group shrinkage of (H, V, D[, A]) (C++-only in the reference, wt.cu:329-336)."""
        self._threshold(self._lib.pdwt_group_soft_threshold, beta, do_threshold_appcoeffs, normalize)

    def shrink(self, beta, do_threshold_appcoeffs=1):
        """shrink(x, t) = x / (1 + t)"""
        rc = self._lib.pdwt_shrink(self._h, float(beta), int(do_threshold_appcoeffs))
        if rc == _lib.ERR_STATE:
            print("Warning: Wavelets(): " + _lib.last_error(self._lib))
            return
        self._check(rc)

    def proj_linf(self, beta, do_threshold_appcoeffs=1):
        """Projection onto the L-infinity ball of radius beta (C++-only in the reference, wt.cu:349-356)."""
        rc = self._lib.pdwt_proj_linf(self._h, float(beta), int(do_threshold_appcoeffs))
        if rc == _lib.ERR_STATE:
            print("Warning: Wavelets(): " + _lib.last_error(self._lib))
            return
        self._check(rc)

    def norm1(self):
        """L1 norm of all the wavelet coefficients."""
        out = self._lib.pdwt_real()
        self._check(self._lib.pdwt_norm1(self._h, C.byref(out)))
        return out.value

    def norm2sq(self):
        """Squared L2 norm of all the wavelet coefficients."""
        out = self._lib.pdwt_real()
        self._check(self._lib.pdwt_norm2sq(self._h, C.byref(out)))
        return out.value

    def _norms_slot(self, out):
        if out is not None:
            dev = _device_array(out, np.float64)
            if dev is None or int(np.prod(dev[1])) < 2:
                raise ValueError("norms: `out` must be a device array of two float64")
            return C.c_void_p(dev[0]), out
        ptr = C.c_void_p()
        self._check(self._lib.pdwt_norms_slot(self._h, C.byref(ptr)))
        return C.c_void_p(None), DeviceArray(self, ptr.value, (2,), np.float64, self._stream())

    def norms_device(self, out=None):
        """NEW: (sum |c|, sum c^2) of all the coefficients as two float64 ON THE DEVICE, enqueued on the plan's stream and not
        waited for (norm1 / norm2sq above block and copy one float to the host each).  Returns a two-element device array:
        ``out`` if given (any ``__cuda_array_interface__`` holder of two float64), else a view of the plan's own slot."""
        arg, view = self._norms_slot(out)
        self._check(self._lib.pdwt_norms_async(self._h, arg))
        return view

    def soft_threshold_norms(self, beta, do_threshold_appcoeffs=0, normalize=0, out=None):
        """NEW: ``soft_threshold(beta, ...)`` and ``norms_device()`` of the result in ONE sweep over the coefficients (the inner
        loop of iterative shrinkage: threshold, then the l1 norm of what is left); same arguments as ``soft_threshold``."""
        arg, view = self._norms_slot(out)
        self._check(self._lib.pdwt_soft_threshold_norms_async(self._h, float(beta), int(do_threshold_appcoeffs), int(normalize), arg))
        return view

    def read_norms(self, view=None):
        """The two float64 a ``norms_device`` / ``soft_threshold_norms`` call left on the device, copied to the host (waits for
        the plan's stream): (sum |c|, sum c^2)."""
        return _read_norms(self._lib, self._h, view)

    def add_wavelet(self, W, alpha=1.0):
        """coefficients += alpha * W.coefficients"""
        rc = self._lib.pdwt_add_wavelet(self._h, W._h, float(alpha))
        if rc != 0:  # the reference prints the reason and carries on (wt.cu:625-650)
            print(("WARNING: " if rc > 0 else "ERROR: ") + _lib.last_error(self._lib))
        return rc

    def set_coeff(self, coeff, num, check=False):
        """Set coefficient band ``num`` (see coeff_only for the numbering); host or device array."""
        dev = _device_array(coeff, self._dtype)
        if dev is not None:  # wt.cu:435-466 with mem_is_on_device = 1
            n = self._check(int(self._lib.pdwt_coeff_count(self._h, int(num), None, None)))
            if int(np.prod(dev[1])) != n:
                raise ValueError("set_coeff: expected %d elements for coefficient %d, got %d" % (n, num, int(np.prod(dev[1]))))
            self._order_after_producer(dev)
            self._check(self._lib.pdwt_set_coeff(self._h, C.c_void_p(dev[0]), int(num), 1))
            return
        coeff = self._checkarray(np.asarray(coeff))
        if check:
            dcoeff = self.coeff_only(num)
            if dcoeff.shape != coeff.shape:
                raise ValueError("set_coefInvalid coefficient shape : expected %s, got %s"
                                 % (str(dcoeff.shape), str(coeff.shape)))
        rows, cols = C.c_int(), C.c_int()
        n = self._lib.pdwt_coeff_count(self._h, int(num), C.byref(rows), C.byref(cols))
        self._check(int(n))
        if coeff.size != n:  # the reference copies blindly (wt.cu:435 "There are no memory check !")
            raise ValueError("set_coeff: expected %d elements for coefficient %d, got %d" % (n, num, coeff.size))
        self._check(self._lib.pdwt_set_coeff(self._h, _ptr(coeff), int(num), 0))

    def set_wavelets_filters(self, filter_name, lowpass, highpass, i_lowpass, i_highpass, LH=None, HL=None,
                             i_LH=None, i_HL=None):
        """
        Set a custom filter bank. This re-defines the current wavelet transform.
        Separable plans use (lowpass, highpass, i_lowpass, i_highpass); non-separable plans
        additionally need the 2D LH, HL, i_LH, i_HL banks (lowpass = LL, highpass = HH).
        """
        arrs = [lowpass, highpass, i_lowpass, i_highpass, LH, HL, i_LH, i_HL]
        if any(len(a) != len(lowpass) for a in arrs if a is not None):
            raise ValueError("All filters must have the same length")
        f = [None if a is None else self._checkarray(np.asarray(a)) for a in arrs]
        name = filter_name.encode("ASCII")
        flen = int(len(lowpass))
        null = C.cast(None, C.POINTER(self._lib.pdwt_real))
        if self.do_separable:
            self._check(self._lib.pdwt_set_filters_forward(self._h, name, flen, self._fptr(f[0]), self._fptr(f[1]), null, null))
            self._check(self._lib.pdwt_set_filters_inverse(self._h, self._fptr(f[2]), self._fptr(f[3]), null, null))
        else:
            if LH is None or HL is None or i_LH is None or i_HL is None:
                raise ValueError("Expected LH and HL filters for non-separable transform")
            # argument order of the C side: (LL, LH, HL, HH)  (src/pypwt.pyx:557-575)
            self._check(self._lib.pdwt_set_filters_forward(self._h, name, flen, self._fptr(f[0]), self._fptr(f[4]), self._fptr(f[5]),
                                                     self._fptr(f[1])))
            self._check(self._lib.pdwt_set_filters_inverse(self._h, self._fptr(f[2]), self._fptr(f[6]), self._fptr(f[7]), self._fptr(f[3])))
        self.hlen = flen
        self.wname = filter_name

    def image_int_ptr(self):
        """Address of the device image (borrowed; valid while the instance lives)."""
        return int(self._lib.pdwt_image_ptr(self._h))

    def coeff_int_ptr(self, num):
        """Address of device coefficient band ``num``."""
        return int(self._lib.pdwt_coeff_ptr(self._h, int(num)))

    def _stream(self):
        return int(self._lib.pdwt_get_stream(self._h) or 0)

    @property
    def image_device(self):
        """NEW: zero-copy device view of the image (``__cuda_array_interface__``); see DeviceArray."""
        return DeviceArray(self, self.image_int_ptr(), (self.Nr, self.Nc) if self.ndim == 2 or self.batched1d else (self.Nc,),
                           self._dtype, self._stream())

    def coeff_device(self, num):
        """NEW: zero-copy device view of coefficient band ``num`` (numbering of coeff_only)."""
        rows, cols = C.c_int(), C.c_int()
        self._check(int(self._lib.pdwt_coeff_count(self._h, int(num), C.byref(rows), C.byref(cols))))
        ptr = self.coeff_int_ptr(num)
        if not ptr:
            raise _lib.PdwtError("coeff_device(%d): %s" % (num, _lib.last_error(self._lib)))
        return DeviceArray(self, ptr, (rows.value, cols.value), self._dtype, self._stream())

    @property
    def coeffs_device(self):
        """NEW: the ``coeffs`` list as zero-copy device views: [A, [H1, V1, D1], ...] (2D) / [A, D1, ...] (1D)."""
        out = [self.coeff_device(0)]
        two_d = (self.ndim == 2) and not self.batched1d
        for lvl in range(self.levels):
            out.append([self.coeff_device(1 + 3 * lvl + k) for k in range(3)] if two_d else self.coeff_device(1 + lvl))
        return out

    @property
    def current_shift(self):
        """(rows, cols) circular shift the last forward() applied (do_cycle_spinning; the reference keeps it in
        Wavelets::current_shift_r/c, pdwt/src/wt.h:27-28, without exposing it to Python)."""
        sr, sc = C.c_int(), C.c_int()
        self._check(self._lib.pdwt_current_shift(self._h, C.byref(sr), C.byref(sc)))
        return int(sr.value), int(sc.value)

    def synchronize(self):
        """Wait for every kernel enqueued by this instance (new; the reference syncs implicitly)."""
        self._check(self._lib.pdwt_synchronize(self._h))

    def cleanup(self):  # should not be called manually
        if getattr(self, "_h", None):
            self._lib.pdwt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass

    @classmethod
    def version(cls):
        """Version of the pypwt API this class is a drop-in for."""
        return "1.0.3"


class Wavelets64(Wavelets):
    """`Wavelets` over the fp64 build of the library (libpypwt_amd_f64.so): float64 images and
    coefficients, same methods.  The reference offers double precision only as a compile-time variant of
    its C++ library (pdwt/Makefile DOUBLEPRECISION, pdwt/src/filters.h:16-30); its Python class is
    float32-only.  Of the tuned kernels this build has the register kernels (2D DWT levels with filters of at
    most 8 taps, 1D DWT level triples, fused 2-tap SWT groups) and the small-image pyramid; everything else runs
    through the generic kernels."""
    _dtype = np.float64
    _variant = "f64"


class BatchedWavelets(object):
    """NEW (not in the reference): one plan over a batch of independent images [B][Nr][Nc].

    Every call (forward / inverse / soft_threshold) processes all B images in the same kernel
    launches; this is the unit that shards across GPUs in bench.py (one plan per rank, no
    collective on the data path).  The image can be generated on the device
    (``fill_hash``) so large batches never cross PCIe.
    """

    _dtype = np.float32
    _variant = "f32"
    _single = Wavelets   # the one-image class of the same build (array checks)

    def __init__(self, batch, Nr, Nc, wname, levels, do_swt=0, ndim=2, device=-1, stream=None, img=None):
        self._lib = _lib.load(self._variant)
        self._h = None
        h = handle_t()
        ptr = None
        if img is not None:
            img = self._single._checkarray(np.asarray(img), (batch, Nr, Nc))
            ptr = _ptr(img)
        rc = self._lib.pdwt_create_batched(ptr, int(batch), int(Nr), int(Nc), wname.encode("ASCII"), int(levels),
                                           1, 1, 0, int(do_swt), int(ndim), int(device),
                                           C.c_void_p(stream) if stream else None, C.byref(h))
        check(rc, "BatchedWavelets()")
        self._h = h
        info, b = PdwtInfo(), C.c_int()
        check(self._lib.pdwt_get_info(self._h, C.byref(info), None, None, None, C.byref(b)))
        self.batch, self.Nr, self.Nc = int(b.value), int(info.Nr), int(info.Nc)
        self.levels, self.hlen, self.ndims, self.do_swt = int(info.nlevels), int(info.hlen), int(info.ndims), int(info.do_swt)
        self.nbands = 3 * self.levels + 1 if self.ndims == 2 else self.levels + 1

    def fill_hash(self, seed, scale=255.0, index_offset=0):
        check(self._lib.pdwt_fill_image_hash(self._h, seed & 0xFFFFFFFF, float(scale), int(index_offset)))

    def forward(self):
        check(self._lib.pdwt_forward(self._h), "forward")

    def inverse(self):
        check(self._lib.pdwt_inverse(self._h), "inverse")

    def soft_threshold(self, beta, do_threshold_appcoeffs=0, normalize=0):
        check(self._lib.pdwt_soft_threshold(self._h, float(beta), int(do_threshold_appcoeffs), int(normalize)))

    def norms_device(self):
        """(sum |c|, sum c^2) over the whole batch into the plan's device slot; not waited for (see Wavelets.norms_device)."""
        check(self._lib.pdwt_norms_async(self._h, None))

    def soft_threshold_norms(self, beta, do_threshold_appcoeffs=0, normalize=0):
        """soft_threshold and the norms of the result in one sweep (see Wavelets.soft_threshold_norms)."""
        check(self._lib.pdwt_soft_threshold_norms_async(self._h, float(beta), int(do_threshold_appcoeffs), int(normalize), None))

    def read_norms(self):
        return _read_norms(self._lib, self._h, None)

    def set_image(self, img):
        img = self._single._checkarray(np.asarray(img), (self.batch, self.Nr, self.Nc))
        check(self._lib.pdwt_set_image(self._h, _ptr(img), 0))

    def coeff(self, num):
        rows, cols = C.c_int(), C.c_int()
        n = check(int(self._lib.pdwt_coeff_count(self._h, int(num), C.byref(rows), C.byref(cols))))
        out = np.zeros((self.batch, rows.value, cols.value), dtype=self._dtype)
        got = self._lib.pdwt_get_coeff(self._h, _ptr(out), int(num))
        if got != n:
            raise RuntimeError("BatchedWavelets.coeff(%d): expected %d, got %d (%s)" % (num, n, got, _lib.last_error(self._lib)))
        return out

    def coeff_at(self, num, image_index):
        """Sub-band `num` of ONE image of the batch (a 128-image shard does not fit a host array per band)."""
        rows, cols = C.c_int(), C.c_int()
        check(int(self._lib.pdwt_coeff_count(self._h, int(num), C.byref(rows), C.byref(cols))))
        out = np.zeros((rows.value, cols.value), dtype=self._dtype)
        got = self._lib.pdwt_get_coeff_at(self._h, _ptr(out), int(num), int(image_index))
        if got != out.size:
            raise RuntimeError("BatchedWavelets.coeff_at(%d, %d): expected %d, got %d (%s)"
                               % (num, image_index, out.size, got, _lib.last_error(self._lib)))
        return out

    def image_at(self, image_index):
        out = np.zeros((self.Nr, self.Nc), dtype=self._dtype)
        got = self._lib.pdwt_get_image_at(self._h, _ptr(out), int(image_index))
        if got != out.size:
            raise RuntimeError("BatchedWavelets.image_at(%d): expected %d, got %d (%s)"
                               % (image_index, out.size, got, _lib.last_error(self._lib)))
        return out

    def norm2sq(self):
        out = self._lib.pdwt_real()
        check(self._lib.pdwt_norm2sq(self._h, C.byref(out)))
        return float(out.value)

    @property
    def image(self):
        out = np.zeros((self.batch, self.Nr, self.Nc), dtype=self._dtype)
        got = self._lib.pdwt_get_image(self._h, _ptr(out))
        if got != out.size:
            raise RuntimeError("BatchedWavelets.image: expected %d, got %d (%s)" % (out.size, got, _lib.last_error(self._lib)))
        return out

    def synchronize(self):
        check(self._lib.pdwt_synchronize(self._h))

    def set_stream(self, stream_ptr):
        check(self._lib.pdwt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def schedule(self):
        """The plan's launch lists ("fwd: ...", "inv: ..."), decided once at creation (plan.cpp: build_schedule)."""
        buf = C.create_string_buffer(4096)
        check(self._lib.pdwt_schedule_string(self._h, buf, len(buf)))
        return buf.value.decode()

    def enable_kernel_timing(self, on=True):
        check(self._lib.pdwt_enable_kernel_timing(self._h, 1 if on else 0))

    def reset_kernel_times(self):
        check(self._lib.pdwt_reset_kernel_times(self._h))

    def time_level(self, level, inverse=False, reps=50):
        """Mean microseconds of the level-`level` launch: `reps` launches back to back between two HIP
        events on the plan's stream (pdwt_time_level)."""
        ms = C.c_float()
        check(self._lib.pdwt_time_level(self._h, int(level), 1 if inverse else 0, int(reps), C.byref(ms)))
        return ms.value * 1e3

    def time_copy(self, elems, reps=50):
        """Mean microseconds of a plain 16-B grid-stride copy of `elems` values out of the plan's image buffer
        (pdwt_time_copy): the measured ceiling for a streaming kernel that reads and writes that many values."""
        ms = C.c_float()
        check(self._lib.pdwt_time_copy(self._h, int(elems), int(reps), C.byref(ms)))
        return ms.value * 1e3

    def copy_capacity(self):
        """largest element count time_copy copies without clamping"""
        return int(self._lib.pdwt_copy_capacity(self._h))

    def kernel_times(self, cap=4096):
        """[(name, milliseconds)] of every launch recorded since the last reset."""
        ms = (C.c_float * cap)()
        names = ((C.c_char * 48) * cap)()
        n = check(self._lib.pdwt_kernel_times(self._h, ms, C.cast(names, C.c_void_p), cap))
        return [(names[i].value.decode(), float(ms[i])) for i in range(min(n, cap))]

    def kernel_families(self, cap=4096):
        """Which of a step's alternative kernels served every recorded launch ("tile", "wave", "ring", "generic", "" for steps with one
        kernel), in the order of kernel_times()."""
        fam = ((C.c_char * 16) * cap)()
        n = check(self._lib.pdwt_kernel_families(self._h, C.cast(fam, C.c_void_p), cap))
        return [fam[i].value.decode() for i in range(min(n, cap))]

    def cleanup(self):
        if getattr(self, "_h", None):
            self._lib.pdwt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


class BatchedWavelets64(BatchedWavelets):
    """`BatchedWavelets` over the fp64 build of the library (float64 images and coefficients)."""
    _dtype = np.float64
    _variant = "f64"
    _single = Wavelets64
