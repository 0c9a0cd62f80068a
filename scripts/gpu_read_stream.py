"""HBM read-stream ceiling (ssimu2_measure_read_stream) for 1, 2 and 4 GiB buffers, three runs each."""
import sys; sys.path.insert(0,'.')
import oavif_amd
with oavif_amd.Ssimu2(0) as s:
    for nb in (1<<30, 2<<30, 4<<30):
        print(nb>>20, "MiB", [round(s.measure_read_stream(nb, 10),1) for _ in range(3)])
