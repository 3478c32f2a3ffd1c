# Host-side AddressSanitizer build of the C-ABI layer (api.hip: argument checks, workspace
# carving, descriptor handling, RCCL binding; layout.hip: geometry, buffer carving, the
# argument checks of the layout builder) linked against the regular kernel objects.
# CPU only -- tests/test_host.py loads it under LD_PRELOAD=$(ASAN_RT) and walks the error
# paths; GPU sanitizer runs are not available on the pool.
ASAN_RT ?= $(shell $(HIPCC) -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null)
ASAN_LIB = ../libspmf_hip_asan.so
%_asan.o: %.hip common.h kernels.h ../../include/spmf_hip.h
	$(HIPCC) $(CXXFLAGS) -O1 -g -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -c $< -o $@
# instrumented: the two files with host-side logic behind the C-ABI (api.hip, layout.hip)
ASAN_OBJS = api_asan.o layout_asan.o $(filter-out api.o layout.o,$(OBJS))
$(ASAN_LIB): $(ASAN_OBJS) exports.map
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -fsanitize=address -shared-libsan -Wl,--version-script=exports.map $(ASAN_OBJS) -o $@ -ldl
asan: $(ASAN_LIB)
	@echo $(ASAN_RT)
