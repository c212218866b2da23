ls /sys/class/drm/ 2>&1 | head; for f in /sys/class/drm/card*/device/hwmon/hwmon*/*; do echo "$f: $(cat $f 2>/dev/null | head -c 80)"; done 2>&1 | head -60
time rocm-smi --showpower --showclocks --json 2>&1 | head -c 1500
echo; which amd-smi; ls /sys/class/drm/card*/device/ | head -80
for f in pp_dpm_sclk gpu_busy_percent mem_busy_percent; do for c in /sys/class/drm/card*/device/$f; do echo $c; cat $c 2>&1 | head -12; done; done
