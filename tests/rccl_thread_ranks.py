"""Helper of test_collective_control_flow_with_thread_ranks (run as a child process, never collected by pytest):
N host threads act as N ranks of mbx_init_broadcast / mbx_comm_agree on ONE device, with MBX_RCCL_LIBRARY pointing at
tests/fake_rccl.c's library (set by the parent, together with the fault-injection variables).  Prints one JSON object."""
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n, root = int(sys.argv[1]), int(sys.argv[2])
    import mbelib_neo_amd as mbx
    from mbelib_neo_amd import _native

    L = _native.lib()
    blob = bytes(mbx.load_tables_blob())
    ident = C.create_string_buffer(128)
    rc = L.mbx_comm_unique_id(ident)
    assert rc == 0, (rc, L.mbx_last_error())
    out = [dict() for _ in range(n)]

    def rank_main(r):
        o = out[r]
        try:
            comm = C.c_void_p()
            o["init"] = L.mbx_comm_init(C.byref(comm), n, ident, r, 0)   # every rank on device 0
            if o["init"] < 0:
                o["error"] = L.mbx_last_error().decode()
                return
            buf = C.create_string_buffer(blob if r == root else bytes(len(blob)), len(blob))
            minmax = (C.c_uint32 * 2)()
            o["bcast"] = L.mbx_init_broadcast(comm, root, 0, buf, len(blob), minmax, None)
            o["bcast_error"] = L.mbx_last_error().decode() if o["bcast"] < 0 else ""
            o["blob_ok"] = buf.raw == blob
            o["minmax"] = (minmax[0], minmax[1])
            o["agree_same"] = L.mbx_comm_agree(comm, 7, minmax, None)
            o["agree_diff"] = L.mbx_comm_agree(comm, 8 if r == n - 1 else 7, minmax, None)
            o["agree_diff_minmax"] = (minmax[0], minmax[1])
            o["destroy"] = L.mbx_comm_destroy(comm)
        except Exception as e:   # noqa: BLE001
            o["exception"] = repr(e)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    stuck = [t.is_alive() for t in threads]
    print(json.dumps({"ranks": out, "stuck": stuck, "checksum": int(L.mbx_table_checksum()) if not any(stuck) else 0}), flush=True)
    if any(stuck):
        os._exit(3)   # threads blocked in a collective cannot be joined


if __name__ == "__main__":
    main()
