package org.apache.singa;

import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.MemorySegment;

/** A host or GPU device of the native library. */
public final class Device implements AutoCloseable {
    final MemorySegment handle;

    private Device(MemorySegment handle) {
        this.handle = handle;
    }

    public static Device host() {
        try {
            return new Device(Singa.checked((MemorySegment) Singa.fn("singa_device_host",
                    FunctionDescriptor.of(Singa.PTR)).invokeExact()));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    public static Device gpu(int id) {
        try {
            return new Device(Singa.checked((MemorySegment) Singa.fn("singa_device_gpu",
                    FunctionDescriptor.of(Singa.PTR, Singa.I32)).invokeExact(id)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** The GPU as seen by the machine's OpenCL runtime (see {@link Singa#numOpenclDevices()}). */
    public static Device opencl(int id) {
        try {
            return new Device(Singa.checked((MemorySegment) Singa.fn("singa_device_opencl",
                    FunctionDescriptor.of(Singa.PTR, Singa.I32)).invokeExact(id)));
        } catch (IllegalStateException e) {
            throw e;
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    /** Wait until the device has finished the work issued so far. */
    public void sync() {
        try {
            Singa.fn("singa_device_sync", FunctionDescriptor.ofVoid(Singa.PTR)).invokeExact(handle);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }

    @Override
    public void close() {
        try {
            Singa.fn("singa_device_free", FunctionDescriptor.ofVoid(Singa.PTR)).invokeExact(handle);
        } catch (Throwable t) {
            throw new IllegalStateException(t);
        }
    }
}
