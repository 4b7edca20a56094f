#!/bin/bash
# Development aid: after a GPU memory fault the runtime leaves gpucore.<pid> in the working directory; this prints, for the
# waves that faulted, the program counter, the instructions around it and the scalar registers (rocgdb, batch mode).
#   tools/lab/gpucore_report.sh gpucore.1234 > gpurun_out/gpucore.txt
core=${1:?usage: gpucore_report.sh gpucore.<pid>}
exec /opt/rocm/bin/rocgdb -batch -q \
  -ex "set pagination off" -ex "set width 0" \
  -ex "info agents" \
  -ex "info threads" \
  -ex "thread apply all -q -s x/6i \$pc-8" \
  -ex "thread apply all -q -s info registers pc exec status trapsts mode m0 vcc" \
  /usr/bin/python3 -c "$core"
