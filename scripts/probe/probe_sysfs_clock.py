"""What can a non-root process read about the GPU's clock and power on the GPU box without starting another program?"""
import glob, os, time
for pat in ("/sys/class/drm/card*/device/pp_dpm_sclk", "/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input",
            "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input",
            "/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input", "/sys/class/drm/card*/device/gpu_busy_percent",
            "/sys/class/kfd/kfd/topology/nodes/*/properties", "/sys/class/drm/card*/device/pp_dpm_mclk"):
    for f in sorted(glob.glob(pat))[:3]:
        try:
            txt = open(f).read().strip().replace("\n", " | ")[:300]
        except Exception as e:
            txt = f"unreadable: {e}"
        print(f, "->", txt)
import shutil
print("rocm-smi", shutil.which("rocm-smi"), "amd-smi", shutil.which("amd-smi"))
try:
    import amdsmi
    print("amdsmi importable")
except Exception as e:
    print("amdsmi not importable:", e)
t0 = time.time(); os.system("rocm-smi --showclocks --showpower 2>&1 | head -30"); print("rocm-smi took", time.time() - t0)
