import ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svi_mapper_amd as s
lib=s.load_library()
for tile in (48,96):
    out=[]
    for stop in (5,1,2,3,4,0):
        ms=C.c_double(0)
        rc=lib.svi_debug_chol_probe(0,tile,300,stop,C.cast(C.byref(ms), C.POINTER(C.c_double)))
        out.append("%d:%.2fus"%(stop,ms.value*1e3))
    print(tile, " ".join(out))
    for mode,name in ((6,"full sweep"),):
        arr=(C.c_double*8)()
        lib.svi_debug_chol_probe(0,tile,50,mode,arr)
        cyc,ticks=arr[0],arr[1]
        print("   %-30s %.0f shader cycles, %.2f us, clock %.2f GHz" % (name, cyc, ticks*0.01, cyc/(ticks*10.0) if ticks else 0))
