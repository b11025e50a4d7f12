#!/bin/bash
# What the GPU box gives a process: CPU quota, allowed CPUs, SMT layout.  Run on the box.
echo "nproc: $(nproc)  online: $(cat /sys/devices/system/cpu/online)"
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "allowed: $(grep Cpus_allowed_list /proc/self/status)"
echo "siblings of cpu0: $(cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list)  cpu1: $(cat /sys/devices/system/cpu/cpu1/topology/thread_siblings_list)"
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA" | head -12
cat /proc/loadavg
