#!/usr/bin/env python3
"""What a process can learn about its GPUs WITHOUT initialising HIP (so that a rank can pin itself to the cores near
its GPU before torch / the HIP runtime start any thread), against what the runtime then reports.  Prints: the KFD
topology nodes (GPU nodes: simd_count > 0) with their PCI address (domain, location_id), render minor, and whether
/dev/dri/renderD<minor> can be opened by this process; the AMD display-class PCI functions of sysfs with their NUMA
node and local_cpulist; *_VISIBLE_DEVICES; then ssimu2_query_device for every HIP device.  Round 5: the `collective`
record showed a rank pinned to cpus 0-15 while its GPU hangs off NUMA node 1 -- sysfs lists every GPU of the host,
the container sees one, and no *_VISIBLE_DEVICES variable says which."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def props(path):
    out = {}
    try:
        for ln in open(path):
            k, _, v = ln.strip().partition(" ")
            out[k] = v
    except Exception as e:
        out["error"] = str(e)
    return out


def main():
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        print(var, "=", os.environ.get(var))
    nodes = sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p)))
    print("kfd nodes:", len(nodes))
    for n in nodes:
        p = props(os.path.join(n, "properties"))
        if int(p.get("simd_count", "0") or 0) <= 0:
            continue
        loc = int(p.get("location_id", "0"))
        dom = int(p.get("domain", "0"))
        addr = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        minor = p.get("drm_render_minor")
        dev = f"/dev/dri/renderD{minor}"
        openable = None
        try:
            fd = os.open(dev, os.O_RDWR)
            os.close(fd)
            openable = True
        except Exception as e:
            openable = f"no ({type(e).__name__}: {e.errno if hasattr(e, 'errno') else ''})"
        numa = None
        try:
            numa = open(f"/sys/bus/pci/devices/{addr}/numa_node").read().strip()
        except Exception:
            pass
        print(f"  node {os.path.basename(n)}: pci {addr} render minor {minor} exists {os.path.exists(dev)} openable {openable} "
              f"numa {numa} gfx_target_version {p.get('gfx_target_version')} unique_id {p.get('unique_id')}")
    from oavif_amd import hostinfo
    print("sysfs AMD display-class functions (PCI address order):")
    for dev in sorted(glob.glob("/sys/bus/pci/devices/*")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            cls = open(os.path.join(dev, "class")).read().strip()
            if not cls.startswith(("0x03", "0x12")):
                continue
            print("  ", os.path.basename(dev), cls, "numa", open(os.path.join(dev, "numa_node")).read().strip(),
                  "local_cpulist", open(os.path.join(dev, "local_cpulist")).read().strip())
        except Exception as e:
            print("  ", dev, "unreadable", e)
    for nd in sorted(glob.glob("/sys/devices/system/node/node*")):
        try:
            print("  ", os.path.basename(nd), "cpulist", open(os.path.join(nd, "cpulist")).read().strip())
        except Exception:
            pass
    print("allowed cpus:", hostinfo.format_cpus(hostinfo.allowed_cpus()), "quota", hostinfo.cgroup_cpu_quota())
    print("node_core_sets(1):", [hostinfo.format_cpus(s) for s in hostinfo.node_core_sets(1)])
    if hasattr(hostinfo, "visible_gpu_pci_addresses"):
        print("hostinfo.visible_gpu_pci_addresses():", hostinfo.visible_gpu_pci_addresses())
    import oavif_amd
    import torch
    for i in range(torch.cuda.device_count()):
        print("HIP device", i, oavif_amd.query_device(i))


if __name__ == "__main__":
    main()
