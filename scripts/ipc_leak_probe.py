"""Does exporting a device buffer through hipIpc keep its memory after hipFree?  One process: 9 rounds of
{hipMalloc 2 MiB, (hipIpcGetMemHandle), hipFree}; the device's free memory after round 1 vs after round 9.
    python scripts/ipc_leak_probe.py
Measured on MI355X / ROCm 7.2 (HSA_ENABLE_IPC_MODE_LEGACY=0): plain alloc + free loses nothing, with the export every
round loses its 2 MiB although no peer ever opened the handle.  The diagnostic behind gaib_comm's pool of exported
buffers (csrc/comm.hip)."""
import ctypes as C

hip = C.CDLL("libamdhip64.so")
SZ = 2 << 20


def free_mem():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value


def main():
    hip.hipSetDevice(0)
    for export in (False, True):
        base = None
        for k in range(9):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(SZ)) == 0
            if export:
                h = (C.c_char * 64)()
                assert hip.hipIpcGetMemHandle(C.byref(h), p) == 0
            assert hip.hipFree(p) == 0
            hip.hipDeviceSynchronize()
            if k == 0:
                base = free_mem()
        print(f"{'export + free' if export else 'alloc + free only'}: {(base - free_mem()) / 2**20:.1f} MiB lost over 8 rounds of 2 MiB", flush=True)


if __name__ == "__main__":
    main()
