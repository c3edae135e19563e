# A/B of two builds of the register-resident wave kernel on the same box: usage scripts/ab_wave.sh variant_lib [batch]
for i in 1 2 3; do
  echo "base:    $(python scripts/stamps_wave.py ${2:-1024} 2>&1 | tail -1 | grep -o '[0-9.]* us per call')"
  echo "variant: $(LEXLS_HIP_LIB=$1 python scripts/stamps_wave.py ${2:-1024} 2>&1 | tail -1 | grep -o '[0-9.]* us per call')"
done
