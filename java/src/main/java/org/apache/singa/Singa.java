package org.apache.singa;

import java.lang.foreign.Arena;
import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.foreign.ValueLayout;
import java.lang.invoke.MethodHandle;
import java.util.concurrent.ConcurrentHashMap;

/**
 * Entry points of the native library, bound once through the foreign
 * function interface.  The library path comes from the system property
 * {@code singa.lib} or the environment variable {@code SINGA_LIB}.
 */
public final class Singa {
    static final Linker LINKER = Linker.nativeLinker();
    static final SymbolLookup LIB = SymbolLookup.libraryLookup(libraryPath(), Arena.global());
    private static final ConcurrentHashMap<String, MethodHandle> CACHE = new ConcurrentHashMap<>();

    static final ValueLayout.OfLong I64 = ValueLayout.JAVA_LONG;
    static final ValueLayout.OfInt I32 = ValueLayout.JAVA_INT;
    static final ValueLayout.OfFloat F32 = ValueLayout.JAVA_FLOAT;
    static final ValueLayout PTR = ValueLayout.ADDRESS;

    private Singa() {}

    private static String libraryPath() {
        String p = System.getProperty("singa.lib");
        if (p == null || p.isEmpty()) p = System.getenv("SINGA_LIB");
        if (p == null || p.isEmpty()) p = "libsinga_amd_cpp.so";
        return p;
    }

    static MethodHandle fn(String name, FunctionDescriptor fd) {
        return CACHE.computeIfAbsent(name, n -> LINKER.downcallHandle(
                LIB.find(n).orElseThrow(() -> new UnsatisfiedLinkError(n)), fd));
    }

    static String lastError() {
        try {
            MemorySegment s = (MemorySegment) fn("singa_last_error",
                    FunctionDescriptor.of(PTR)).invokeExact();
            return s.reinterpret(4096).getString(0);
        } catch (Throwable t) {
            return t.toString();
        }
    }

    static MemorySegment checked(MemorySegment handle) {
        if (handle.equals(MemorySegment.NULL)) throw new IllegalStateException(lastError());
        return handle;
    }

    static void checked(int rc) {
        if (rc != 0) throw new IllegalStateException(lastError());
    }

    public static String version() {
        try {
            MemorySegment s = (MemorySegment) fn("singa_version",
                    FunctionDescriptor.of(PTR)).invokeExact();
            return s.reinterpret(64).getString(0);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static int numGpus() {
        try {
            return (int) fn("singa_num_gpus", FunctionDescriptor.of(I32)).invokeExact();
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static int numOpenclDevices() {
        try {
            return (int) fn("singa_num_opencl_devices", FunctionDescriptor.of(I32)).invokeExact();
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** Element-wise function of one tensor: abs exp log relu sigmoid sign sqrt square tanh. */
    public static Tensor unary(String op, Tensor x) {
        try (Arena a = Arena.ofConfined()) {
            return new Tensor(checked((MemorySegment) fn("singa_unary",
                    FunctionDescriptor.of(PTR, PTR, PTR)).invokeExact(a.allocateFrom(op), x.handle)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** add, sub, mul or div of two tensors (broadcast like the native library). */
    public static Tensor binary(String op, Tensor x, Tensor y) {
        try (Arena a = Arena.ofConfined()) {
            return new Tensor(checked((MemorySegment) fn("singa_binary",
                    FunctionDescriptor.of(PTR, PTR, PTR, PTR))
                    .invokeExact(a.allocateFrom(op), x.handle, y.handle)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static Tensor Square(Tensor x) { return unary("square", x); }
    public static Tensor ReLU(Tensor x) { return unary("relu", x); }
    public static Tensor Exp(Tensor x) { return unary("exp", x); }
    public static Tensor Tanh(Tensor x) { return unary("tanh", x); }

    public static Tensor Mult(Tensor a, Tensor b) {
        try {
            return new Tensor(checked((MemorySegment) fn("singa_matmul",
                    FunctionDescriptor.of(PTR, PTR, PTR)).invokeExact(a.handle, b.handle)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static Tensor Sum(Tensor x, int axis) {
        try {
            return new Tensor(checked((MemorySegment) fn("singa_sum",
                    FunctionDescriptor.of(PTR, PTR, I32)).invokeExact(x.handle, axis)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static Tensor SoftMax(Tensor x) {
        try {
            return new Tensor(checked((MemorySegment) fn("singa_softmax",
                    FunctionDescriptor.of(PTR, PTR)).invokeExact(x.handle)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
