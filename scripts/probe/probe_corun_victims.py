"""Round 6: which packed-f32 instruction pattern returns other bits beside int8-MFMA waves (csrc/debug/dbg_victims.hip; debug library:
APS_LIB_PATH=<pkg>/lib/libaps_hip_dbg.so).  A worker thread runs one known-answer variant in a loop on its stream while the main thread
launches the co-runner of dbg_corun.hip (mode 64 = v_mfma_i32_16x16x64_i8, 8 = v_mfma_f32_32x32x16_f16, 0 = idle, none = no co-runner)."""
import sys, ctypes, threading, time
sys.path.insert(0, ".")
import apsamd
lib = apsamd._capi.lib
lib.aps_dbg_corun.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
lib.aps_dbg_victim.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
NAMES = {0: "pk_mul -> s_nop 0 -> pk_add (dependent; the compiler's sequence)", 1: "pk_mul -> s_nop 1 -> pk_add", 2: "pk_mul -> s_nop 7 -> pk_add",
         3: "pk_mul -> s_nop 0 -> v_add_f32 x2 (plain consumers)", 4: "four independent pk_mul, results read >= 16 wait states later",
         5: "pk_mov_b32 op_sel:[1,0] -> s_nop 0 -> pk_mul", 6: "pk_fma -> pk_fma on the result, back to back", 7: "no packed instruction (control)",
         8: "v_mov_b32 of one half of each pair -> pk_add of the pairs, back to back", 9: "pk_mul op_sel_hi:[1,0] -> negating pk_add -> pk_add neg, one wait state each",
         10: "s_and_b64 exec -> pk_mul under the new mask (other lanes keep a sentinel)", 11: "s_mov_b32 -> pk_mov_b32 from the SGPR pair -> pk_mul",
         12: "exec shrinks, one v_add, s_or_b64 exec back to full -> pk_mul at once", 13: "pk_mul -> global_store_dwordx2 of the pair at once -> read back",
         14: "pk_mov_b32 builds a 64-bit address -> global_load_dword through it at once",
         15: "pk_mul -> v_add_f32 reads its LOW half, NO wait state", 16: "pk_mul -> v_add_f32 reads its HIGH half, NO wait state",
         17: "pk_fma op_sel_hi:[0,1,1] -> v_add_f32 reads its low half, NO wait state", 18: "pk_mul -> s_nop 0 -> v_add_f32 reads its low half",
         19: "pk_mul with an SGPR pair, low half broadcast (op_sel_hi:[1,0]) -> pk_add", 20: "pk_add with the inline constant 2.0 (op_sel_hi:[1,0]) -> pk_mul",
         21: "pk_mul -> pk_add inside a lane-divergent loop (1-8 trips per lane, data-dependent reset)"}
NWG, ITERS, LAUNCHES = 2048, 4000, 12
def run(var, mode):
    tot = [0, 0]
    def work():
        apsamd._capi.check(lib.aps_set_thread_device(0))
        for _ in range(LAUNCHES):
            m = ctypes.c_ulonglong(0)
            apsamd._capi.check(lib.aps_dbg_victim(var, NWG, ITERS, ctypes.byref(m)))
            tot[0] += m.value
            tot[1] += 1
    th = threading.Thread(target=work)
    th.start()
    k = 0
    while th.is_alive():
        if mode is None:
            time.sleep(0.001)
        else:
            apsamd._capi.check(lib.aps_dbg_corun(mode, 4096, 400))
            k += 1
    th.join()
    return tot[0], k
results = 2 * NWG * 256 * ITERS * LAUNCHES
print("results checked per cell: %.2e" % results)
for var in ([int(v) for v in sys.argv[1:]] or range(22)):
    row = []
    for mode in (None, 64, 2):
        bad, k = run(var, mode)
        row.append("%s: %d wrong (%d co-runs)" % ("quiet" if mode is None else "mode %d" % mode, bad, k))
    print("variant %d  %-68s | %s" % (var, NAMES[var], " | ".join(row)), flush=True)
