//go:build cgo && ksolve

// volumeusage_ksolve.go — goes into pkg/scheduling (package scheduling, next to volumeusage.go). The ksolve flattener
// (pkg/controllers/provisioning/scheduling/ksolve_flatten.go) hands the device the volumes in use and the CSINode attach
// limits of every existing node (VolumeUsage.ExceedsLimits / Add, volumeusage.go:193-209); VolumeUsage keeps both in
// unexported fields (volumeusage.go:178-182).
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Go toolchain).
package scheduling

// Tracked returns the volumes in use per CSI driver and the per-driver attach limits (read-only use).
func (v *VolumeUsage) Tracked() (Volumes, map[string]int) {
	if v == nil {
		return nil, nil
	}
	return v.volumes, v.limits
}
