//go:build !amdgolden

package main

import "log"

// Without -tags amdgolden the r900 package has no test hook (go/r900/amd_golden.go), so the
// second-stage golden is skipped; everything else is written.
func r900Golden(golden string, capture []byte) {
	log.Println("r900 second-stage golden skipped: build with -tags amdgolden after copying go/r900/amd_golden.go into r900/")
}
