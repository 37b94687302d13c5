# Builds the C-ABI HIP library (gfx950 only) and the C oracle helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := medmoe_amd/csrc
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
LIB   := medmoe_amd/lib/libmedmoe_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wno-unused-result

all: $(LIB)

build/%.o: $(CSRC)/%.hip $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p medmoe_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
