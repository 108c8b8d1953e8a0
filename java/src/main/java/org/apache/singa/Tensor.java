package org.apache.singa;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.MemorySegment;

/**
 * A tensor of the native library.  Method names follow the classes the
 * reference generates with SWIG (SetFloatValue, GetFloatValue,
 * CopyFloatDataFromHostPtr ...), so code written against them keeps working.
 */
public final class Tensor implements AutoCloseable {
    public static final int FLOAT32 = 0;
    public static final int INT32 = 2;

    final MemorySegment handle;

    Tensor(MemorySegment handle) {
        this.handle = handle;
    }

    public Tensor(long[] shape) {
        this(shape, FLOAT32, null);
    }

    public Tensor(long[] shape, int dtype, Device device) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment s = a.allocateFrom(Singa.I64, shape);
            MemorySegment dev = device == null ? MemorySegment.NULL : device.handle;
            this.handle = Singa.checked((MemorySegment) Singa.fn("singa_tensor_new",
                    FunctionDescriptor.of(Singa.PTR, Singa.PTR, Singa.I32, Singa.I32, Singa.PTR))
                    .invokeExact(s, shape.length, dtype, dev));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public long[] shape() {
        try {
            int n = (int) Singa.fn("singa_tensor_ndim",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR)).invokeExact(handle);
            long[] out = new long[n];
            for (int i = 0; i < n; i++) {
                out[i] = (long) Singa.fn("singa_tensor_dim",
                        FunctionDescriptor.of(Singa.I64, Singa.PTR, Singa.I32)).invokeExact(handle, i);
            }
            return out;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public long Size() {
        try {
            return (long) Singa.fn("singa_tensor_size",
                    FunctionDescriptor.of(Singa.I64, Singa.PTR)).invokeExact(handle);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void SetFloatValue(float v) {
        try {
            Singa.checked((int) Singa.fn("singa_tensor_set_value",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR, Singa.F32)).invokeExact(handle, v));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void CopyFloatDataFromHostPtr(float[] src, long n) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment s = a.allocateFrom(Singa.F32, src);
            Singa.checked((int) Singa.fn("singa_tensor_copy_from_float",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR, Singa.PTR, Singa.I64))
                    .invokeExact(handle, s, n));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void CopyIntDataFromHostPtr(int[] src, long n) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment s = a.allocateFrom(Singa.I32, src);
            Singa.checked((int) Singa.fn("singa_tensor_copy_from_int",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR, Singa.PTR, Singa.I64))
                    .invokeExact(handle, s, n));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void GetFloatValue(float[] dst, long n) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment s = a.allocate(Singa.F32, n);
            Singa.checked((int) Singa.fn("singa_tensor_get_float",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR, Singa.PTR, Singa.I64))
                    .invokeExact(handle, s, n));
            MemorySegment.copy(s, Singa.F32, 0, dst, 0, (int) n);
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void GetIntValue(int[] dst, long n) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment s = a.allocate(Singa.I32, n);
            Singa.checked((int) Singa.fn("singa_tensor_get_int",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR, Singa.PTR, Singa.I64))
                    .invokeExact(handle, s, n));
            MemorySegment.copy(s, Singa.I32, 0, dst, 0, (int) n);
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public Tensor Reshape(long[] shape) {
        try (Arena a = Arena.ofConfined()) {
            MemorySegment s = a.allocateFrom(Singa.I64, shape);
            return new Tensor(Singa.checked((MemorySegment) Singa.fn("singa_tensor_reshape",
                    FunctionDescriptor.of(Singa.PTR, Singa.PTR, Singa.PTR, Singa.I32))
                    .invokeExact(handle, s, shape.length)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public Tensor T() {
        try {
            return new Tensor(Singa.checked((MemorySegment) Singa.fn("singa_tensor_transpose",
                    FunctionDescriptor.of(Singa.PTR, Singa.PTR)).invokeExact(handle)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public void ToDevice(Device d) {
        try {
            Singa.checked((int) Singa.fn("singa_tensor_to_device",
                    FunctionDescriptor.of(Singa.I32, Singa.PTR, Singa.PTR)).invokeExact(handle, d.handle));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    @Override
    public void close() {
        try {
            Singa.fn("singa_tensor_free", FunctionDescriptor.ofVoid(Singa.PTR)).invokeExact(handle);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
